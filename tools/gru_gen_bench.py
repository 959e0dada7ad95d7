import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from musicgeneration_amd.melody_rnn import Event_Melody_RNN
torch.manual_seed(0)
net = Event_Melody_RNN(init_dim=32, event_dim=308, hidden_dim=512, rnn_layers=3, dropout=0.3).cuda().eval()
init = torch.randn(32, 32, device="cuda")
steps = 2048
net.generate(init, 64, greedy=0.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = net.generate(init, steps, greedy=0.0)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"Event_Melody_RNN.generate B=32 steps={steps}: {1e3*dt/steps:.4f} ms/step, {32*steps/dt:,.0f} tokens/s")
