import time, torch
from torch.autograd import Function
dev = torch.device("cuda")
class F(Function):
    @staticmethod
    def forward(ctx, x, a, b, c):
        ctx.save_for_backward(x)
        ctx.a = a
        return x.view_as(x)
    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g, None, None, None
x = torch.ones(1024, 128, device=dev, requires_grad=True)
def step(n=11):
    h = x
    for _ in range(n):
        h = F.apply(h, 1, None, "s")
    l = h[:1, :1].sum()
    l.backward()
for _ in range(50): step()
torch.cuda.synchronize()
for n in (11, 22):
    t0 = time.perf_counter()
    for _ in range(500): step(n)
    torch.cuda.synchronize()
    print(n, "nodes:", (time.perf_counter() - t0) / 500 * 1e6, "us per fwd+bwd")
# the same without the engine
class Ctx:
    def save_for_backward(self, *t): self.saved_tensors = t
def manual(n=11):
    h = x.detach(); tape = []
    for _ in range(n):
        c = Ctx(); h = F.forward(c, h, 1, None, "s"); tape.append(c)
    g = torch.ones_like(h)
    for c in reversed(tape):
        g = F.backward(c, g)[0]
for _ in range(50): manual()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): manual(11)
torch.cuda.synchronize()
print("manual 11:", (time.perf_counter() - t0) / 500 * 1e6, "us")
