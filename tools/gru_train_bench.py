"""Event_Melody_RNN training step (Train + cross-entropy + backward) at the reference's configuration
(Event_MelodyRNN/config.py: hidden 512, 3 layers, batch 100, window 200), next to torch.nn.GRU (MIOpen) on the same GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd.melody_rnn import Event_Melody_RNN
B, T, V, H, NL = 100, 200, 308, 512, 3
torch.manual_seed(0)
net = Event_Melody_RNN(init_dim=32, event_dim=V, hidden_dim=H, rnn_layers=NL, dropout=0.3).cuda().train()
events = torch.randint(0, V, (T, B), device="cuda")
init = torch.randn(B, 32, device="cuda")
lossf = torch.nn.CrossEntropyLoss()
def own():
    out = net.Train(init, events[:-1])
    lossf(out.view(-1, V), events.view(-1)).backward()
def lib():
    hid = net.init_to_hidden(init)
    x = net.event_embedding(torch.cat([net.get_primary_event(B), events[:-1]], 0))
    y, _ = net.rnn(x, hid.contiguous())
    lossf(net.output_fc(y).view(-1, V), events.view(-1)).backward()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
t_own, t_lib = timed(own), timed(lib)
print(f"GRU train step B={B} T={T} H={H} layers={NL}: libmgx {t_own*1e3:.1f} ms ({B*T/t_own:,.0f} events/s) | torch.nn.GRU fp32 (MIOpen) {t_lib*1e3:.1f} ms ({B*T/t_lib:,.0f} events/s)")
