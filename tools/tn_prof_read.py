"""per-shape medians of the two launches of the weight-gradient GEMM from a rocprofv3 --kernel-trace database of tools/tn_prof.py"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
main = [(e - s) / 1e3 for n, s, e in rows if 'gemm_tn_kernel' in n]
red = [(e - s) / 1e3 for n, s, e in rows if 'gemm_tn_reduce' in n]
gap = [(s2 - e1) / 1e3 for (n1, s1, e1), (n2, s2, e2) in zip(rows, rows[1:]) if 'gemm_tn_kernel' in n1 and 'gemm_tn_reduce' in n2]
shapes = ((16384, 2304, 768), (16384, 768, 768), (16384, 3072, 768), (16384, 768, 3072), (5120, 2304, 768), (5120, 3072, 768), (16384, 1536, 768))
med = lambda v: sorted(v)[len(v) // 2]
for i, sh in enumerate(shapes):
    m, r, g = main[20 * i + 5:20 * i + 20], red[20 * i + 5:20 * i + 20], gap[20 * i + 5:20 * i + 20]
    print(sh, "main %.1f us  reduce %.1f us  gap %.1f us  -> %.0f TFLOP/s on the main launch" % (med(m), med(r), med(g), 2.0 * sh[0] * sh[1] * sh[2] / med(m) / 1e6))
