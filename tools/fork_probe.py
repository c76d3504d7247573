"""What a fork / join over two HIP streams costs between kernels (GPU):
[A on s0] -> [B on s1 || C on s0] -> [D on s0] against the same four kernels on
one stream.  The kernels are fills, so running B and C side by side gains
nothing: the difference is the price of the two cross-stream waits."""
import torch

dev = torch.device("cuda:0")
a = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
b = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
c = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
d = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def serial():
    with torch.cuda.stream(s0):
        a.fill_(1); b.fill_(2); c.fill_(3); d.fill_(4)


def forked():
    with torch.cuda.stream(s0):
        a.fill_(1)
        e0 = torch.cuda.Event(); e0.record(s0)
    with torch.cuda.stream(s1):
        s1.wait_event(e0)
        b.fill_(2)
        e1 = torch.cuda.Event(); e1.record(s1)
    with torch.cuda.stream(s0):
        c.fill_(3)
        s0.wait_event(e1)
        d.fill_(4)


for name, fn in (("serial", serial), ("forked", forked), ("serial", serial), ("forked", forked)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(s0)
    for _ in range(50):
        fn()
    t1.record(s0)
    torch.cuda.synchronize()
    print(f"{name}: {t0.elapsed_time(t1) / 50 * 1e3:.1f} us per round")
