import csv, collections, sys, glob
for d in sys.argv[1:]:
    for f in glob.glob(f'{d}/*/*counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('vj::','').replace('(CascadeArgs)','')[:52]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in agg.items():
            if 'cascade' in k: print(k, {a: f'{b:.4g}' for a, b in v.items()})
