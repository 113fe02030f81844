# Build-container only (imports /root/reference): the unchanged reference vs itself under a different BLAS thread count.
import sys, os
sys.dont_write_bytecode = True
sys.argv = ["x"]
sys.path.insert(0, "/root/repo/tests/golden"); sys.path.insert(0, "/root/repo")
import make_golden as G
import numpy as np, torch
O = G.O
NSTEP = 300
ttr = dict(O.synth_table(16 * NSTEP, 1, snr=0.15, quant="bf16"))
ttr["vlogit"] = np.zeros((16 * NSTEP, 60), np.float32); ttr["slogit"] = ttr["vlogit"]
conf = np.array(G.CONFS["c4"])
args = G.mkargs(inner_representation_size=128, batchnorm=True, drpt=0.0, epochs=1, batchsize=16)
def run(nt):
    torch.set_num_threads(nt)
    model = G.ntu.Searchable_Skeleton_Image_Net(args, conf)
    G.load_det(model, conf, args, 77)
    opt = torch.optim.Adam(model.central_params(), lr=args.eta_max, weight_decay=1e-4)
    sched = G.sc.LRCosineAnnealingScheduler(args.eta_max, args.eta_min, args.Ti, args.Tm, 625.0)
    crit = torch.nn.CrossEntropyLoss(); model.train(True); out = []
    for data in G.ListLoader(ttr, 16):
        opt.zero_grad(); o = model((data["rgb"], data["ske"])); loss = crit(o, data["label"])
        sched.step(); sched.update_optimizer(opt); loss.backward(); opt.step(); out.append(loss.item())
    return np.array(out)
a, b = run(8), run(1)
for s in (0, 1, 2, 5, 10, 20, 40, 80, 160, 250, 299): print(s, round(a[s], 5), round(b[s], 5), round(b[s] - a[s], 5))
print("mean 8 threads", a.mean(), "mean 1 thread", b.mean())
