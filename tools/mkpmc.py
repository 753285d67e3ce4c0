"""Builds profiles/rNN_pmc_traffic.json from two rocprofv3 PMC passes over `bench.py --steps 1 --warmup 1`
(one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE): HBM bytes per C-ABI entry-point call, by bench.py's profile tag.
    python tools/mkpmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>
FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 B, MI355X_MICROARCH.md, HBM section); WRITE_SIZE is
used as reported.  Counters report KiB."""
import csv, glob, json, re, sys, collections

# (tag, kernels of the entry point, kernels whose launches count its calls - None: every launch is a call)
TAGS = [('fft_cols_f64', ['k_fft_cols<double', 'k_cols2p_first<double', 'k_cols2p_second<double'], ['k_fft_cols<double', 'k_cols2p_first<double']),
        ('fft_rows_f64', ['k_fft_rows<double'], None),
        # (4096-long columns: the column pass of a filtered inverse is two launches, the template path of a scale three)
        ('fft_cols_f32', ['k_fft_cols<float', 'k_template_cols', 'k_template2p_', 'k_cols2p_first<float', 'k_cols2p_second<float'],
         ['k_fft_cols<float', 'k_template_cols', 'k_template2p_first', 'k_cols2p_first<float']),
        ('fft_rows_f32', ['k_fft_rows<float'], None),
        # the pixel-window undo as two 1-D circular filters (nf_axis_filter): one kernel per axis
        ('axis_filter_cols', [r'k_line_filter<\w+, ?(true|1)', 'k_line_filter_cols_fused'], None),
        ('axis_filter_rows', [r'k_line_filter<\w+, ?(false|0)', 'k_line_filter_rows_fused'], None),
        # one nf_noise_map call = (breaks, first scale of a tile and lane only) + block_stats + window_rms0 + block_core + block_classes + windows + paint_blocks
        ('noise_map', ['k_noise_'], ['k_noise_windows'])]


def load(d, counter):
    fn = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if r['Counter_Name'] == counter:
            out[r['Kernel_Name']].append(float(r['Counter_Value']))
    return out


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
res = {}
for tag, pats, callKernel in TAGS:
    names = sorted(k for k in fetch if any(re.search(p, k) for p in pats))
    if not names:
        continue
    if callKernel is None:
        calls = sum(len(fetch[k]) for k in names)
    else:
        calls = sum(len(fetch[k]) for k in names if any(c in k for c in callKernel))
    f = sum(sum(fetch[k]) for k in names) * 1024 * 2 / 1e9 / calls
    w = sum(sum(write.get(k, [])) for k in names) * 1024 / 1e9 / calls
    res[tag] = {'kernels': [k[:90] for k in names], 'calls': calls, 'fetch_GB_corrected': round(f, 4), 'write_GB': round(w, 4),
                'traffic_GB': round(f + w, 4)}
json.dump({'_note': "HBM traffic per C-ABI entry-point call from rocprofv3 PMC passes over `bench.py --steps 1 --warmup 1` on MI355X "
                    "(separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs, summed over the kernels of the entry point and averaged over its "
                    "calls). FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md section HBM); "
                    "WRITE_SIZE is used as reported. Units: GB (1e9 bytes); counters report KiB.", 'kernels': res},
          open(sys.argv[3], 'w'), indent=1)
print(json.dumps(res, indent=1))
