"""Per-dispatch view of one FCN forward from a rocprofv3 rocpd database: grid, duration, and the MFMA work implied by the
grid (block = 16x16 px x NT*32 output channels).  usage: fcn_layers.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,grid_x,grid_y,workgroup_x,lds_size from kernels order by start").fetchall()
prep = [i for i, r in enumerate(rows) if r[0].startswith("lm_k_prepare")]
a, b = prep[-2], prep[-1]
tot = 0
for r in rows[a:b]:
    d = (r[2] - r[1]) / 1e3
    tot += d
    print(f"{r[0][:40]:40s} grid=({r[3]//r[5]:5d},{r[4]:3d}) lds={r[6]:6d} {d:8.1f} us")
print("frame total us", round(tot, 1))
