"""GPU: the N > 1 path with the HIP matcher on every rank.  On a one-GPU box the two ranks share
cuda:0 and reduce over gloo (BC_DIST_BACKEND=gloo); with two or more GPUs the same tests also run
over RCCL ("nccl").  Both product entry points are exercised from a bare shell environment: they
start their own ranks (dist.launch_workers) and rank 0 prints."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _bare_env(backend):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["BC_DIST_BACKEND"] = backend
    return env


def _backends():
    import torch
    return ["gloo"] + (["nccl"] if torch.cuda.device_count() >= 2 else [])


def test_bench_self_launch_two_ranks(gpu_device):
    """`python bench.py --gpus 2` with no launcher in the environment prints one valid line for N = 2 whose
    job-wide matched count equals a single process over the same 2 x n reads of the seeded stream."""
    from barcoder_amd import DeviceLibrary, GpuMatcher, synth as S
    n, G = 1_000_000, 20_000
    lib = S.library(G, 20)
    m = GpuMatcher(DeviceLibrary(lib, 20, gpu_device))
    reads = np.concatenate([S.reads(lib, 20, 2, a, 500_000, 1) for a in range(0, 2 * n, 500_000)])
    want = int(m.match(reads, None, 2, "seed").counts.sum())
    for backend in _backends():
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", str(n),
                              "--guides", str(G), "--steps", "2", "--warmup", "1", "--no-cpu", "--no-secondary"],
                             capture_output=True, text=True, timeout=900, env=_bare_env(backend), cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, out.stdout
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
        assert rec["config"]["job_matched_reads"] == want, backend
        assert backend in rec["config"]["parallelism"]
        assert rec["roofline"]["frac"] > 0 and "cpu_baseline" not in rec


def test_bench_refuses_mismatched_world(gpu_device):
    env = dict(_bare_env("gloo"), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "1000"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)


@pytest.mark.parametrize("scenario,compress,ranks", [("fwd_single", False, 2), ("fwd_single", True, 2), ("paired", False, 2),
                                                     ("paired_swapped", False, 3), ("paired", True, 2)])
def test_heuristicount_cli_two_ranks_equals_reference(gpu_device, tmp_path, scenario, compress, ranks):
    """`python -m barcoder_amd.heuristicount --gpus N`: stdout byte-identical to the reference's single-process
    run — byte-range shards of a plain single-end file, ranges of both mate files that hold the same reads for plain
    paired input (line arithmetic, ingest.paired_read_ranges), chunk striding for compressed input."""
    with gzip.open(os.path.join(GOLDEN, "heuristicount_main.json.gz"), "rt") as f:
        cases = json.load(f)
    s = next((x for x in cases if x["name"] == scenario), None) or next(x for x in cases if x["paired"])
    fa = tmp_path / "lib.fasta"
    fa.write_text("".join(f">bc{i}\n{b}\n" for i, b in enumerate(s["barcodes"])))

    def write(path, reads):
        op = gzip.open if str(path).endswith(".gz") else open
        with op(path, "wt") as f:
            for i, r in enumerate(reads):
                f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n" if s["ext"] == "fastq" else r + "\n")
    ext = s["ext"] + (".gz" if compress else "")
    files = [tmp_path / f"r1.{ext}"]
    write(files[0], s["reads1"])
    if s["paired"]:
        files.append(tmp_path / f"r2.{ext}")
        write(files[1], s["reads2"])
    if scenario == "paired_swapped":  # the mate files in the other order: the sampler finds the swap, same output
        files.reverse()
    for backend in _backends():
        out = subprocess.run([sys.executable, "-m", "barcoder_amd.heuristicount", str(fa)] + [str(p) for p in files] +
                             ["--gpus", str(ranks), "--chunk-reads", "257"], cwd=ROOT, capture_output=True, text=True,
                             timeout=900, env=_bare_env(backend))
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout == s["stdout"], (scenario, compress, backend)
        assert "Total Reads" in out.stderr


def test_long_barcodes_two_ranks_equal_one(gpu_device, tmp_path):
    """Barcodes of more than 32 nt over two ranks (byte-range shards of one plain file, every rank on the device
    ingest path in its text-hash mode, undocumented windows gathered as rows of text): stdout and the undocumented
    table equal the single-process run, which test_heuristicount_gpu.py holds against the oracle."""
    import random
    rng = random.Random(3)
    k = 44
    lib = list(dict.fromkeys("".join(rng.choice("ACGT") for _ in range(k)) for _ in range(300)))
    L, R = "CCGGATTCAG", "GTTTTAGAGC"
    reads = []
    for i in range(20000):
        bc = rng.choice(lib) if rng.random() > 0.15 else "".join(rng.choice("ACGT") for _ in range(k))
        r = "".join(rng.choice("ACGT") for _ in range(6)) + L + bc + R + "".join(rng.choice("ACGT") for _ in range(rng.randrange(5)))
        if rng.random() < 0.02:
            p = rng.randrange(len(r)); r = r[:p] + "N" + r[p + 1:]
        reads.append(r)
    fa = tmp_path / "long.fasta"
    fa.write_text("".join(f">b{i}\n{s}\n" for i, s in enumerate(lib)))
    rd = tmp_path / "long.reads"
    rd.write_text("".join(r + "\n" for r in reads))
    outs = []
    for ranks in (1, 2):
        out = subprocess.run([sys.executable, "-m", "barcoder_amd.heuristicount", str(fa), str(rd), "--gpus", str(ranks)],
                             cwd=ROOT, capture_output=True, text=True, timeout=900, env=_bare_env("gloo"))
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append((out.stdout, [ln for ln in out.stderr.splitlines() if "Barcode" in ln or "Fraction" in ln or ln.rstrip().endswith(("*  1", "*  2"))]))
    assert outs[0][0] == outs[1][0] and len(outs[0][0].splitlines()) > 250
    assert outs[0][1] == outs[1][1] and any("Undocumented" in ln for ln in outs[0][1])


@pytest.mark.parametrize("layout", ["paired", "gz"])
def test_long_barcodes_stride_plan_with_idle_ranks(gpu_device, tmp_path, layout):
    """Barcodes of more than 32 nt on the plans that split by CHUNK — mate pairs (host front half, chunk c -> rank
    c % world) and one compressed file (device ingest, bc_stream_count_file_part) — over three ranks with FEWER chunks
    than ranks: a rank that owns no chunk contributes an empty undocumented table, which must have the text-row shape
    of the others (round 3's advisor: it came back as packed words and broke the merge) — stdout and the summary equal
    the single-process run."""
    import random
    rng = random.Random(7)
    k = 40
    lib = list(dict.fromkeys("".join(rng.choice("ACGT") for _ in range(k)) for _ in range(200)))
    L, R = "CCGGATTCAG", "GTTTTAGAGC"
    comp = str.maketrans("ACGT", "TGCA")
    r1, r2 = [], []
    for i in range(3000):
        bc = rng.choice(lib) if rng.random() > 0.2 else "".join(rng.choice("ACGT") for _ in range(k))
        mid = L + bc + R
        r1.append("".join(rng.choice("ACGT") for _ in range(4)) + mid + "AC")
        r2.append("GT" + mid.translate(comp)[::-1] + "".join(rng.choice("ACGT") for _ in range(3)))
    fa = tmp_path / "long.fasta"
    fa.write_text("".join(f">b{i}\n{s}\n" for i, s in enumerate(lib)))
    if layout == "paired":
        files = [tmp_path / "m1.reads", tmp_path / "m2.reads"]
        files[0].write_text("".join(r + "\n" for r in r1))
        files[1].write_text("".join(r + "\n" for r in r2))
    else:
        files = [tmp_path / "one.reads.gz"]
        with gzip.open(files[0], "wt") as f:
            f.write("".join(r + "\n" for r in r1))
    outs = []
    for ranks in (1, 3):
        out = subprocess.run([sys.executable, "-m", "barcoder_amd.heuristicount", str(fa)] + [str(p) for p in files] +
                             ["--gpus", str(ranks), "--chunk-reads", "2000"], cwd=ROOT, capture_output=True, text=True,
                             timeout=900, env=_bare_env("gloo"))
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append((out.stdout, [ln for ln in out.stderr.splitlines() if "Barcode" in ln or "Fraction" in ln]))
    assert outs[0] == outs[1] and len(outs[0][0].splitlines()) > 150


_RCCL_ONE_RANK = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["BC_ROOT"])
import torch, torch.distributed as dist
from barcoder_amd import dist as D, synth as S, DeviceLibrary, GpuMatcher, MultiGpuMatcher
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["BC_PORT"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
G = 5000
c = torch.arange(G, dtype=torch.int64, device="cuda")
f = torch.full((G,), D.I64_MAX, dtype=torch.int64, device="cuda")
f[::3] = 7
c2, f2 = D.reduce_counts(c.clone(), f.clone(), device=0)                      # RCCL SUM / MIN of int64, device tensors
assert torch.equal(c2, c) and torch.equal(f2, f)
c3, f3, pend = D.reduce_counts(c.clone(), f.clone(), device=0, async_op=True)   # the bench's overlapped form
pend.wait(); torch.cuda.synchronize()
assert torch.equal(c3, c) and torch.equal(f3, f)
cn, fn = D.reduce_counts(np.arange(G, dtype=np.uint64), np.full(G, D.U64_MAX, np.uint64), device=0)  # numpy in, numpy out
assert cn.tolist() == list(range(G)) and int(fn.min()) == D.I64_MAX
dist.barrier(); dist.destroy_process_group()
# the one-process form: ncclCommInitAll + grouped ncclAllReduce (uint64 SUM, MIN) on a communicator of one
lib = S.library(3000, 20)
reads = S.reads(lib, 20, 2, 0, 40_001, 1)
want = GpuMatcher(DeviceLibrary(lib, 20, 0)).match(reads, None, 2, "seed", first_index=5)
mm = MultiGpuMatcher(lib, 20, [0])
assert mm.uses_rccl, "BC_MULTI_RCCL=1 must give a communicator"
got = mm.match(reads, None, 2, "seed", first_index=5)
for a in ("best_idx", "best_dist", "n_best", "counts", "first_seen"):
    assert np.array_equal(getattr(got, a), getattr(want, a)), a
mm.close()
print("rccl-one-rank-ok")
"""


def test_rccl_branch_on_a_group_of_one(gpu_device, tmp_path):
    """The RCCL code paths, executed on this box's one GPU in a group of one rank: torch.distributed's "nccl" backend
    under dist.reduce_counts (device tensors, sync and async, and the numpy form), and bc_multi_gpu_match's own
    communicator (BC_MULTI_RCCL=1).  A group of one moves no data between devices — what this pins is that librccl
    loads, the communicator initialises on gfx950, and the collectives accept our dtypes, ops and streams; the job-wide
    arithmetic over several ranks is pinned by the gloo tests above.  Own process: the group must not outlive the test."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rccl_one.py"
    script.write_text(_RCCL_ONE_RANK)
    env = dict(_bare_env("nccl"), BC_ROOT=ROOT, BC_PORT=str(port), BC_MULTI_RCCL="1")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "rccl-one-rank-ok" in out.stdout, (out.stdout + out.stderr)[-3000:]
