"""print a rocprofv3 *kernel_stats.csv with short kernel names: python tools/kstats.py <dir> [filter]"""
import csv, glob, re, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(srl_\w+|k_\w+|radix_sort_\w+|DeviceSelect\w*|scan\w*|__amd_rocclr_\w+|run_length\w*|lookback_scan\w*|partition\w*)", r["Name"])
        name = m.group(1) if m else r["Name"][:50]
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        print(f"{name:34s} calls {r['Calls']:>4s}  total {float(r['TotalDurationNs']) / 1e3:10.1f} us  avg {float(r['AverageNs']) / 1e3:9.1f} us  max {float(r['MaxNs']) / 1e3:9.1f} us")
