"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/.
usage: python tools/summarize_profiles.py <tag> <round-prefix, e.g. r01> [name, default bench_bs32; bench_c3 / bench_c5 for the other configs]
Writes profiles/<prefix>_<name>_kernel_stats.csv (copy of rocprofv3's kernel_stats) and
profiles/<prefix>_<name>_pmc_hbm.csv (HBM bytes per launch: FETCH_SIZE in KiB, doubled on gfx950 as
MI355X_MICROARCH.md prescribes (the counter sees 64 B of each 128-B request), + WRITE_SIZE in KiB)."""
import csv
import glob
import shutil
import sys
from collections import defaultdict

tag, prefix = sys.argv[1], sys.argv[2]
name = sys.argv[3] if len(sys.argv) > 3 else 'bench_bs32'
base = f'gpurun_out/prof_{tag}'
stats = glob.glob(f'{base}/stats/*/*kernel_stats.csv')[0]
shutil.copy(stats, f'profiles/{prefix}_{name}_kernel_stats.csv')


def per_launch(path, counter):
    acc, n = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        acc[r['Kernel_Name']] += float(r['Counter_Value'])
        n[r['Kernel_Name']].add(r['Dispatch_Id'])
    return {k: (acc[k] / len(n[k]), len(n[k])) for k in acc}


fetch = per_launch(glob.glob(f'{base}/fetch/*/*counter_collection.csv')[0], 'FETCH_SIZE')
write = per_launch(glob.glob(f'{base}/write/*/*counter_collection.csv')[0], 'WRITE_SIZE')
rows = []
for k, (f_kb, nl) in fetch.items():
    w_kb = write.get(k, (0.0, 0))[0]
    fb, wb = f_kb * 1024 * 2, w_kb * 1024
    rows.append((k, nl, f_kb, fb, w_kb, wb, fb + wb))
rows.sort(key=lambda r: -r[6] * r[1])
with open(f'profiles/{prefix}_{name}_pmc_hbm.csv', 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['Kernel_Name', 'launches', 'FETCH_SIZE_KB_per_launch_raw', 'FETCH_bytes_per_launch_corrected_x2', 'WRITE_SIZE_KB_per_launch',
                'WRITE_bytes_per_launch', 'HBM_bytes_per_launch'])
    for r in rows:
        w.writerow([r[0], r[1], f'{r[2]:.1f}', int(r[3]), f'{r[4]:.1f}', int(r[5]), int(r[6])])
fused = [r for r in rows if any(n in r[0] for n in ('k_ffn_xr', 'k_ffn_xs', 'k_ffn_x32', 'k_ffn_fused', 'k_ffn_strip'))]   # the fused FFN forward, all variants
if fused:
    tot = sum(r[6] * r[1] for r in fused) / sum(r[1] for r in fused)
    print(f'fused FFN forward (k_ffn_xs + k_ffn_x32), all variants averaged: {int(tot)} HBM bytes per launch  (what bench.py reads as roofline.traffic)')
for r in rows[:12]:
    print(f'{r[0][:70]:70s} n={r[1]:4d}  {r[6] / 1e6:9.1f} MB/launch')
