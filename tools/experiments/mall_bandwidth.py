import torch, time
dev = torch.device("cuda")
def bw(nbytes, reps=20):
    x = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): s = x.sum()
    e1.record(); torch.cuda.synchronize()
    t_read = e0.elapsed_time(e1) / reps
    e0.record()
    for _ in range(reps): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    t_copy = e0.elapsed_time(e1) / reps
    print(f"{nbytes/2**20:8.0f} MiB: repeated read {nbytes/t_read/1e9:7.2f} TB/s   copy (r+w) {2*nbytes/t_copy/1e9:7.2f} TB/s")
for mb in (16, 32, 64, 128, 192, 256, 512, 1024, 2048):
    bw(mb * 2**20)
