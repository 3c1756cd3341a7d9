"""Average PMC counter values per kernel from rocprofv3 --pmc CSV output.
Usage: python tools/pmc_summary.py <dir-or-counter_collection.csv> [...]"""
import csv, sys, os, collections

def summarize(path):
    if os.path.isdir(path):
        path = [os.path.join(path, f) for f in os.listdir(path) if f.endswith('counter_collection.csv')][0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as fh:
        for row in csv.DictReader(fh):
            acc[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
    for kern, ctrs in acc.items():
        print('##', kern[:90])
        for name, vals in sorted(ctrs.items()):
            print('   %-28s n=%-4d avg=%.4g' % (name, len(vals), sum(vals)/len(vals)))

for p in sys.argv[1:]:
    summarize(p)
