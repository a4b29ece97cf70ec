import torch as th, time
x = th.rand(8,16,2048,2048, device="cuda")
def timeit(f, n=10):
    f(); th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); th.cuda.synchronize()
    return e0.elapsed_time(e1)/n
flat = x.view(-1)
print("vector_norm", timeit(lambda: th.linalg.vector_norm(x)))
print("dot", timeit(lambda: th.dot(flat, flat)))
print("sum", timeit(lambda: x.sum()))
print("square.sum", timeit(lambda: x.square().sum()))
print("mul scalar", timeit(lambda: x * 0.5))
m = th.rand(8,1,2048,2048, device="cuda") > 0.4
print("where", timeit(lambda: th.where(m, x, 0.0)))
print("mul mask", timeit(lambda: x * m))
print("masked_fill", timeit(lambda: x.masked_fill(~m, 0.0)))
