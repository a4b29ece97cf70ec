"""ATen reductions for the user-side loss term mean(x^2) of bench.py (one pass over the [8,16,2048,2048] image): which spelling streams fastest."""
import torch as th
x = th.rand(8, 16, 2048, 2048, device="cuda")
def t(fn, reps=20):
    fn(); th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): r = fn()
    e1.record(); th.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, float(r)
n = x.numel()
cands = {
 "vector_norm(x)^2/n": lambda: th.linalg.vector_norm(x).square() / n,
 "vector_norm(x.view(8,-1),dim=1)": lambda: th.linalg.vector_norm(x.view(8, -1), dim=1).square().sum() / n,
 "vector_norm(x.view(128,-1),dim=1)": lambda: th.linalg.vector_norm(x.view(128, -1), dim=1).square().sum() / n,
 "vector_norm(x.view(2048,-1),dim=1)": lambda: th.linalg.vector_norm(x.view(2048, -1), dim=1).square().sum() / n,
 "vector_norm(x.view(-1,2048),dim=1)": lambda: th.linalg.vector_norm(x.view(-1, 2048), dim=1).square().sum() / n,
 "dot": lambda: th.dot(x.view(-1), x.view(-1)) / n,
 "x.square().sum()": lambda: x.square().sum() / n,
 "x.pow(2).mean()": lambda: x.pow(2).mean(),
 "sum(view(16384,-1),1)": lambda: x.view(16384, -1).square_().sum() if False else th.linalg.vector_norm(x.view(16384, -1), dim=1).square().sum() / n,
}
for k, f in cands.items():
    ms, v = t(f)
    print(f"{k:40s} {ms:7.3f} ms  {2.147/ms:5.2f} TB/s  value {v:.7f}")
