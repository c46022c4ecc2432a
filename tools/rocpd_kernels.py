"""Per-kernel summary (count, mean / min / max microseconds, workgroups) of a rocprofv3 --kernel-trace database."""
import glob, sqlite3, sys
path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
f = path if path.endswith(".db") else sorted(glob.glob(path + "/**/*.db", recursive=True))[-1]
db = sqlite3.connect(f)
q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, max(d.end-d.start)/1000.0, "
     "d.grid_size_x*d.grid_size_y*d.grid_size_z/(d.workgroup_size_x*d.workgroup_size_y*d.workgroup_size_z), sum(d.end-d.start)/1000.0 "
     "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name, 6 order by 7 desc limit ?")
print(f"{'kernel':58s} {'n':>5s} {'mean_us':>9s} {'min_us':>9s} {'max_us':>9s} {'wgs':>6s} {'total_us':>10s}")
for r in db.execute(q, (top,)):
    name = r[0].replace("_ZN12_GLOBAL__N_1", "")[:58]
    print(f"{name:58s} {r[1]:5d} {r[2]:9.1f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:6d} {r[6]:10.1f}")
