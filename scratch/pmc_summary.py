import csv, collections, sys, glob
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            if 'rdf_fast' in r['Kernel_Name'] or 'rdf_tile' in r['Kernel_Name']:
                key = (r['Kernel_Name'].split('(')[0][-40:], r['Grid_Size'])
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
                agg[key]['dur_ms'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
        for k, v in agg.items():
            print(d, k)
            for c, vals in sorted(v.items()):
                print('   %-28s %.4g  (n=%d)' % (c, sum(vals) / len(vals), len(vals)))
