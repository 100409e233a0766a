"""Per-launch durations of the panel kernel for the last solve in a rocprofv3 --kernel-trace csv."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
pan = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_panel', 'k_merge', 'k_corner'))]
n = per or len(pan) // 3
last = pan[-n:]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000 for r in last]
g = [int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // 256 for r in last]
print('n', n, 'sum', sum(d))
print('dur', ' '.join('%.1f' % x for x in d))
print('wgs', ' '.join(str(x) for x in g))
