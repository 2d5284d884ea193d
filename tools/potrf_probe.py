"""Two ranks on ONE GPU factorise the same K + 1e-6 I (SE kernel, n = 2000) repeatedly: which launch conditions make
rocSOLVER potrf report a failed minor?  (diagnostic of the failure seen in the 2-rank gloo rehearsal)"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, '/root/repo')
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1: dist.init_process_group("gloo")
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
rng = np.random.default_rng(1002)
n, d = 2000, 8
Z = torch.tensor(rng.uniform(size=(n, d)), dtype=torch.float64, device=dev)
ls = torch.tensor(np.exp(rng.uniform(np.log(0.7), np.log(3.0), size=d)), dtype=torch.float64, device=dev)
A = Z / ls
d2 = (A * A).sum(-1)[:, None] + (A * A).sum(-1)[None, :] - 2.0 * A @ A.T
K = 0.7921 * torch.exp(-0.5 * d2.clamp_min(0.0)) + 1e-6 * torch.eye(n, dtype=torch.float64, device=dev)
K = 0.5 * (K + K.T)
torch.cuda.synchronize()
Kh = K.cpu().numpy()
Lref = np.linalg.cholesky(Kh)
print(f"[rank {rank}] host LAPACK factorises it: min diag(L) = {Lref.diagonal().min():.3e}", flush=True)
if world > 1: dist.barrier()
side = torch.cuda.Stream(device=dev)
def run(tag, fn, reps=8):
  fails, infos, diffs = 0, [], []
  for r in range(reps):
    L, info = fn()
    torch.cuda.synchronize()
    i = int(info.item())
    if i != 0: fails += 1; infos.append(i)
    else: diffs.append(float(np.abs(L.cpu().numpy() - Lref).max()))
  print(f"[rank {rank}] {tag}: {fails}/{reps} failed, info values {infos}, max |L - L_lapack| of the good ones {max(diffs) if diffs else None}", flush=True)
def on_default():
  return torch.linalg.cholesky_ex(K)
def on_side():
  side.wait_stream(torch.cuda.current_stream(dev))
  with torch.cuda.stream(side):
    out = torch.linalg.cholesky_ex(K)
  torch.cuda.current_stream(dev).wait_stream(side)
  return out
def default_synced():
  torch.cuda.synchronize()
  out = torch.linalg.cholesky_ex(K)
  torch.cuda.synchronize()
  return out
def blocked(nb=250):
  # right-looking blocked Cholesky out of small potrf calls (n <= 250: rocSOLVER's unblocked kernel), trsm and gemm
  Lm = K.clone(); info = torch.zeros((), dtype=torch.int32, device=dev)
  for j in range(0, n, nb):
    e = min(j + nb, n)
    Ljj, inf = torch.linalg.cholesky_ex(Lm[j:e, j:e])
    info = torch.maximum(info, (inf != 0).to(torch.int32) * (inf + j).to(torch.int32))
    Lm[j:e, j:e] = Ljj
    if e < n:
      Lm[e:, j:e] = torch.linalg.solve_triangular(Ljj, Lm[e:, j:e].T, upper=False).T
      Lm[e:, e:] -= Lm[e:, j:e] @ Lm[e:, j:e].T
  return torch.tril(Lm), info
for tag, fn in (("default stream", on_default), ("side stream", on_side), ("default + device sync around", default_synced), ("blocked (nb=250) from small potrf + trsm + gemm", blocked)):
  run(tag, fn)
  if world > 1: dist.barrier()
if world > 1: dist.destroy_process_group()
