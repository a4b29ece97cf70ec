import torch as th
x = th.empty(8,16,2048,2048, device="cuda")
y = th.rand(8,16,2048,2048, device="cuda")
def timeit(f, n=10):
    f(); th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); th.cuda.synchronize()
    return e0.elapsed_time(e1)/n
t = timeit(lambda: x.fill_(1.5)); print("fill_ 2.1GB", t, "ms", 2.147/t, "TB/s")
t = timeit(lambda: x.zero_()); print("zero_", t, 2.147/t)
t = timeit(lambda: x.copy_(y)); print("copy_", t, 4.295/t, "TB/s r+w")
t = timeit(lambda: y.sum()); print("sum (read)", t, 2.147/t)
