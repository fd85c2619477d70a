"""Host cost of PyTorch's stream primitives (per call, microseconds)."""
import time
import torch
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
s = torch.cuda.Stream(device=dev)
cur = torch.cuda.current_stream(dev)
N = 5000


def t(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    dt = (time.perf_counter() - t0) / N * 1e6
    torch.cuda.synchronize()
    print("%-42s %6.2f us" % (name, dt))


def ctx():
    with torch.cuda.stream(s):
        pass


def setstream():
    torch.cuda.set_stream(s)
    torch.cuda.set_stream(cur)


def ev_pair():
    e = torch.cuda.Event()
    e.record(cur)
    s.wait_event(e)


t("torch.cuda.current_stream(dev)", lambda: torch.cuda.current_stream(dev))
t("with torch.cuda.stream(s): pass", ctx)
t("set_stream(s); set_stream(cur)", setstream)
t("Event(); record(cur); s.wait_event", ev_pair)
t("s.wait_stream(cur)", lambda: s.wait_stream(cur))
x = torch.zeros(16, device=dev)
t("x.record_stream(s)", lambda: x.record_stream(s))
t("torch.empty((100, 452))", lambda: torch.empty((100, 452), device=dev))
t("s.cuda_stream", lambda: s.cuda_stream)
