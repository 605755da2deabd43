"""dev tool: learned whitening at descriptor size D on N vectors / n pairs: GPU (whiten_learn.hip) vs the numpy oracle.
usage: tools/whiten_learn_bench.py [D N npairs] [--cpu]"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from gandtr_amd import whiten_learn

args = [a for a in sys.argv[1:] if not a.startswith("--")]
d, n, npairs = (int(v) for v in args[:3]) if len(args) >= 3 else (2048, 60000, 20000)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
basis = torch.randn(d, d, generator=g, device=dev) * torch.linspace(1.5, 0.2, d, device=dev)[None, :]
base = torch.randn(n // 2, d, generator=g, device=dev) @ basis.t()
X = torch.cat([base, base + 0.3 * torch.randn(n // 2, d, generator=g, device=dev) @ basis.t()])
X = torch.nn.functional.normalize(X, dim=1)
q = torch.randint(0, n // 2, (npairs,), generator=torch.Generator().manual_seed(1))
p = q + n // 2
torch.cuda.synchronize()
t0 = time.perf_counter()
m, P, info = whiten_learn.whitenlearn(X.t(), q, p, return_info=True)
torch.cuda.synchronize()
t_gpu = time.perf_counter() - t0
t0 = time.perf_counter()
m, P, info = whiten_learn.whitenlearn(X.t(), q, p, return_info=True)
torch.cuda.synchronize()
t_gpu2 = time.perf_counter() - t0
# sign-free check of the result at full size: P S P^T = I
Xd = X.double()
diff = Xd[q.to(dev)] - Xd[p.to(dev)]
S = diff.t() @ diff / npairs
err = float((P @ S @ P.t() - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())
out = {"workload": "whitenlearn D=%d N=%d pairs=%d (float64)" % (d, n, npairs), "gpu_s_first": round(t_gpu, 3), "gpu_s": round(t_gpu2, 3),
       "jacobi_sweeps": info["jacobi_sweeps"], "jitter_steps": info["cholesky_jitter_steps"], "max|P S P^T - I|": err,
       "gemm_gflop_fp64": round(2.0 * d * d * (n + npairs) / 1e9 + 3 * 2.0 * d ** 3 / 1e9, 1)}
if "--cpu" in sys.argv:
    from oracle import whiten_oracle as W
    Xn = X.t().double().cpu().numpy()
    t0 = time.perf_counter()
    m_ref, P_ref, w_ref = W.whitenlearn(Xn, q.numpy(), p.numpy())
    out["cpu_numpy_s"] = round(time.perf_counter() - t0, 2)
    out["cpu_threads"] = os.cpu_count()
    out["rows_up_to_sign_vs_numpy"] = W.rows_up_to_sign(P.cpu().numpy(), P_ref)
print(json.dumps(out))
