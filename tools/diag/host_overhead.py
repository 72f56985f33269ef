"""Host cost per step of the pieces the reference's training loop dictates (diagnostic; GPU): stock torch.optim.Adam on the 14
parameter tensors of M2, zero_grad, an autograd Function round trip, loss.item()."""
import time, torch
dev = "cuda"
shapes = [(128, 1026), (128,), (128, 128), (128,), (16, 128), (16,), (16, 128), (16,), (128, 529), (128,), (128, 128), (128,), (513, 128), (513,)]
params = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.05) for s in shapes]
opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.999))
flatg = torch.randn(sum(p.numel() for p in params), device=dev)
def set_grads():
    o = 0
    for p in params:
        p.grad = flatg[o:o + p.numel()].view_as(p); o += p.numel()
def timeit(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    host = (time.perf_counter() - t) / n
    torch.cuda.synchronize(); tot = (time.perf_counter() - t) / n
    return host * 1e6, tot * 1e6
def adam():
    set_grads(); opt.step(); opt.zero_grad()
print("Adam(foreach default).step + zero_grad + 14 grad view assignments: host %.0f us, wall %.0f us" % timeit(adam))
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *ps):
        ctx.save_for_backward(x); return x * 1.0
    @staticmethod
    def backward(ctx, g):
        return (g,) + (None,) * 14
x = torch.randn(8192, 513, device=dev, requires_grad=False)
def fb():
    y = F.apply(x, *params); l = y.sum(); l.backward()
print("Function.apply(+14 params) + sum + backward: host %.0f us, wall %.0f us" % timeit(fb))
l = torch.zeros((), device=dev)
print("3x .item(): host %.0f us" % timeit(lambda: (l.item(), l.item(), l.item()))[0])
opt2 = torch.optim.Adam(params, lr=1e-4, fused=True)
def adam2():
    set_grads(); opt2.step(); opt2.zero_grad()
print("Adam(fused=True) for comparison: host %.0f us, wall %.0f us" % timeit(adam2))
