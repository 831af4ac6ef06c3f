"""Prints per-dispatch durations of selected kernels from a rocprofv3 kernel_trace.csv."""
import csv, sys, glob
pat = sys.argv[2] if len(sys.argv) > 2 else "flood_explore"
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if pat in r["Kernel_Name"]]
for r in sel[-int(sys.argv[3]) if len(sys.argv) > 3 else -12:]:
    print(r["Kernel_Name"][:60], r["Grid_Size_X"], "%.1f us" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), "vgpr", r["VGPR_Count"], "lds", r["LDS_Block_Size"])
