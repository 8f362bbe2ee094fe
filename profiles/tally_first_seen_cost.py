"""What the first-seen half of the tally costs: the pass (matcher + tally) timed with per-guide counts AND first-seen,
with counts only (k_hist_part then issues one LDS atomic per record instead of two, k_first_refine has nothing to do),
and without any tally.  usage: python3 profiles/tally_first_seen_cost.py [guides]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from barcoder_amd import DeviceLibrary, GpuMatcher, synth as S

G = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000
n, K = 100_000_000, 20
dev = torch.device("cuda", 0)
dl = DeviceLibrary(S.library(G, K), K, 0)
m = GpuMatcher(dl)
st = torch.cuda.current_stream().cuda_stream
reads = torch.empty(n, dtype=torch.int64, device=dev)
bi = torch.empty(n, dtype=torch.int32, device=dev)
bd = torch.empty(n, dtype=torch.uint8, device=dev)
nb = torch.empty(n, dtype=torch.uint8, device=dev)
c32 = torch.zeros(G, dtype=torch.int32, device=dev)
fs = torch.full((G,), -1, dtype=torch.int64, device=dev)
m.synth_reads_dev(2, 0, n, 1, reads.data_ptr(), st)
torch.cuda.synchronize()
def t(**kw):
    return m.time_dev(reads.data_ptr(), n, bi.data_ptr(), 2, "seed", iters=60, d_best_dist=bd.data_ptr(), d_n_best=nb.data_ptr(), stream=st, **kw)
for rep in range(2):
    a = t()
    b = t(d_counts=c32.data_ptr())
    c = t(d_counts=c32.data_ptr(), d_first_seen=fs.data_ptr())
    print(f"G={G} rep {rep}: kernel alone {a:.4f} ms, + counts {b:.4f} ms, + counts and first-seen {c:.4f} ms")
