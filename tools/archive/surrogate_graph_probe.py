"""Feasibility probe: one Adam step of the surrogate (Linear(3,100)+Sigmoid -> LSTM(100,100) over L <= 4 cells -> Linear(100,1) ->
Sigmoid, MSE) on the GPU as a replayed HIP graph (manual LSTM cell, padded bucket + mask) vs PyTorch-CPU eager."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mfas_amd.search import surrogate as S

dev = torch.device("cuda:0")
NMAX, L, n = 1024, 4, 300
torch.manual_seed(0)
cpu = S.SimpleRecurrentSurrogate(100, 3, 100)
x = torch.randint(0, 4, (L, n, 3)).float(); y = torch.rand(n, 1)
opt = torch.optim.Adam(cpu.parameters(), lr=1e-3); crit = torch.nn.MSELoss()
t0 = time.perf_counter()
for _ in range(100):
    opt.zero_grad(); l = crit(cpu(x), y); l.backward(); opt.step()
t_cpu = (time.perf_counter() - t0) / 100

# GPU: parameters as leaf tensors, manual LSTM
torch.manual_seed(0)
ref = S.SimpleRecurrentSurrogate(100, 3, 100)
P = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in ref.named_parameters()}
def fwd(xp):      # xp (L, NMAX, 3)
    h = torch.zeros(NMAX, 100, device=dev); c = torch.zeros(NMAX, 100, device=dev)
    for t in range(L):
        e = torch.sigmoid(xp[t] @ P["embedding.0.weight"].t() + P["embedding.0.bias"])
        g = e @ P["lstm.weight_ih_l0"].t() + P["lstm.bias_ih_l0"] + h @ P["lstm.weight_hh_l0"].t() + P["lstm.bias_hh_l0"]
        i, f, gg, o = g.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
    return torch.sigmoid(h @ P["hid2val.weight"].t() + P["hid2val.bias"])
xp = torch.zeros(L, NMAX, 3, device=dev); yp = torch.zeros(NMAX, 1, device=dev); mask = torch.zeros(NMAX, 1, device=dev); inv_n = torch.ones((), device=dev)
xp[:, :n] = x.to(dev); yp[:n] = y.to(dev); mask[:n] = 1; inv_n.fill_(1.0 / n)
gopt = torch.optim.Adam(list(P.values()), lr=1e-3, capturable=True, foreach=True)
def step():
    gopt.zero_grad(set_to_none=False)
    loss = (mask * (fwd(xp) - yp) ** 2).sum() * inv_n
    loss.backward()
    gopt.step()
    return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
t_eager = (time.perf_counter() - t0) / 50
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 200
    print(f"cpu eager {t_cpu*1e3:.2f} ms/step   gpu eager {t_eager*1e3:.2f}   gpu graph replay {t_graph*1e3:.3f} ms/step  loss {float(out):.5f} (cpu loss {float(l):.5f})")
except Exception as e:
    print(f"cpu eager {t_cpu*1e3:.2f} ms/step   gpu eager {t_eager*1e3:.2f}   graph capture failed: {type(e).__name__}: {e}")
