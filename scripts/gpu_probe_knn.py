"""GPU probe: which GEMM/top-k chunking the bulk graph builder should use (harness, not product)."""
import sys, time, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "helix-db_amd")]
import torch
dev = torch.device("cuda")
n, d = 1_000_000, 768
x = torch.randn(n, d, device=dev); x /= x.norm(dim=1, keepdim=True)
xb = x.to(torch.bfloat16)
def t(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.time() - t0) / reps * 1e3
try:
    torch.mm(xb[:8], xb[:8].t(), out_dtype=torch.float32); ok = True
except Exception as e:
    ok = False; print("out_dtype unsupported:", e)
print("bf16->f32 out_dtype:", ok)
for rc, cc in [(2048, 1 << 18), (4096, 1 << 18), (4096, 1 << 20), (8192, 1 << 20)]:
    cc = min(cc, n)
    if ok:
        ms_mm = t(lambda: torch.mm(xb[:rc], xb[:cc].t(), out_dtype=torch.float32))
    else:
        ms_mm = float("nan")
    ms_mm32 = t(lambda: x[:rc] @ x[:cc].t())
    s = x[:rc] @ x[:cc].t()
    ms_tk = t(lambda: torch.topk(s, 81, dim=1))
    per_full = (n / rc) * (n / cc)
    print(f"rows {rc} cols {cc}: mm_bf16 {ms_mm:.2f} ms  mm_f32 {ms_mm32:.2f} ms  topk81 {ms_tk:.2f} ms  -> full kNN est "
          f"{per_full * (min(ms_mm, ms_mm32) if ok else ms_mm32) / 1e3:.1f}s gemm + {per_full * ms_tk / 1e3:.1f}s topk")
    del s
