import torch, time
x = torch.arange(1 << 20, dtype=torch.float32, device="cuda:0")
class Wrap:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2, "strides": None}
w = Wrap(x.data_ptr() + 4 * 16, 1000)
t = torch.as_tensor(w, device="cuda:0")
print("as_tensor ok:", t.shape, t.dtype, float(t[0]), t.data_ptr() == x.data_ptr() + 64)
t[0] = -5.0
print("aliasing:", float(x[16]))
# D2D copy speed of hipMemcpyAsync-like paths
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda:0"); b = torch.empty_like(a)
torch.cuda.synchronize(); t0 = time.perf_counter(); b.copy_(a); torch.cuda.synchronize(); print("torch copy 1 GiB: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
