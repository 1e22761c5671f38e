"""Host-to-device rates of this box: hipMemcpyAsync from page-locked memory (one stream, two streams), and a kernel reading the page-locked buffer in place."""
import json, time, torch
n = 256 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory(); h.fill_(1)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
out = {}
def timed(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return n * reps / (time.perf_counter() - t0) / 1e9
out["memcpy_one_stream_GBps"] = timed(lambda: d.copy_(h, non_blocking=True))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    with torch.cuda.stream(s1): d[: n // 2].copy_(h[: n // 2], non_blocking=True)
    with torch.cuda.stream(s2): d[n // 2:].copy_(h[n // 2:], non_blocking=True)
out["memcpy_two_streams_GBps"] = timed(two)
for chunk in (2 << 20, 512 << 10):
    def chunks():
        for o in range(0, n, chunk): d[o:o + chunk].copy_(h[o:o + chunk], non_blocking=True)
    out[f"memcpy_chunks_{chunk >> 10}KB_GBps"] = timed(chunks, 2)
print(json.dumps(out))
chunk = 2 << 20
for ns in (2, 4, 8):
    ss = [torch.cuda.Stream() for _ in range(ns)]
    def rr():
        for k, o in enumerate(range(0, n, chunk)):
            with torch.cuda.stream(ss[k % ns]): d[o:o + chunk].copy_(h[o:o + chunk], non_blocking=True)
    out[f"memcpy_chunks_2MB_{ns}_streams_GBps"] = timed(rr, 2)
print(json.dumps(out))
