#!/usr/bin/env python3
"""End-to-end timing of barcoder_amd.targets on an E. coli-sized synthetic genome (BASELINE configs[3] scale):
4.64 Mbp circular chromosome, 4 300 genes, guides sampled next to NGG sites on both strands plus random ones.
usage: python profiles/targets_bench.py [n_guides] [mismatches]"""
import io, os, sys, time, tempfile, types
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barcoder_amd import synth as S, targets as T, genbank as GB

n_guides = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
mm = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = 4_641_652
seq = S.genome(L, seed=4)
rng = np.random.default_rng(1)
starts = np.sort(rng.integers(0, L - 3000, 4300))
feats = [GB.GbFeature("gene", [(int(a), int(a + rng.integers(300, 2500)))], int(rng.choice([1, -1])),
                      {"locus_tag": [f"b{i:04d}"], "gene": [f"g{i}"]}) for i, a in enumerate(starts)]
d = tempfile.mkdtemp(prefix="bc_tb_")
gb = os.path.join(d, "genome.gb")
GB.write_genbank(gb, [GB.GbRecord("NC_SYN.1", "NC_SYN", seq, "circular", "Synthetic coli", feats)])
comp = str.maketrans("ACGT", "TGCA")
guides = []
while len(guides) < n_guides * 9 // 10:
    p = int(rng.integers(0, L - 30))
    if seq[p + 21:p + 23] == "GG":
        guides.append(seq[p:p + 20])
    elif seq[p:p + 2] == "CC":
        guides.append(seq[p + 3:p + 23].translate(comp)[::-1])
letters = np.array(list("ACGT"))
guides += ["".join(letters[rng.integers(0, 4, 20)]) for _ in range(n_guides - len(guides))]
fa = os.path.join(d, "guides.fasta")
with open(fa, "w") as f:
    f.write("".join(f">s{i}\n{g}\n" for i, g in enumerate(guides)))
t = {}
t0 = time.perf_counter(); gl = T.read_guides(fa); t["read_guides"] = time.perf_counter() - t0
t0 = time.perf_counter(); chroms, genes, seq_lens, org, topo, cnt = T.chromosomes_and_genes(gb); t["genbank"] = time.perf_counter() - t0
t0 = time.perf_counter(); rows = T.find_hits([(a, b.upper()) for a, b in gl], chroms, pam="NGG", num_mismatches=mm, report_unaligned=True); t["find_hits"] = time.perf_counter() - t0
t0 = time.perf_counter(); rows2 = T.find_hits([(a, b.upper()) for a, b in gl], chroms, pam="NGG", num_mismatches=mm, report_unaligned=True); t["find_hits_again"] = time.perf_counter() - t0
t0 = time.perf_counter(); ann = T.annotate(rows, genes); t["annotate"] = time.perf_counter() - t0
t0 = time.perf_counter(); cols, table = T.final_table(ann, seq_lens); t["final_table"] = time.perf_counter() - t0
t0 = time.perf_counter(); tsv = T.to_tsv(cols, table); t["to_tsv"] = time.perf_counter() - t0
t0 = time.perf_counter(); js = T.to_json(cols, table); t["to_json"] = time.perf_counter() - t0
args = types.SimpleNamespace(sgrna_file=fa, genome_file=gb, pam="NGG", mismatches=mm, pam_direction="downstream", json=False)
buf = io.StringIO()
t0 = time.perf_counter()
err, sys.stderr = sys.stderr, io.StringIO()
try:
    T.main(args, out=buf)
finally:
    sys.stderr = err
t["main_total"] = time.perf_counter() - t0
print({"guides": n_guides, "mismatches": mm, "hit_rows": len(rows), "table_rows": len(table), "tsv_bytes": len(tsv),
       "seconds": {k: round(v, 3) for k, v in t.items()}})
import shutil; shutil.rmtree(d, ignore_errors=True)
