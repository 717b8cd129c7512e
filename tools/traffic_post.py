"""Turn gpurun_out/traffic/traffic_raw.json (tools/prof_traffic.sh) into the per-kernel HBM traffic
summary committed under profiles/: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports
half of a coalesced streaming read (guides/MI355X_MICROARCH.md, HBM section) -> x2, calibrated in the
same run on a 1 GiB copy with our own kernels; WRITE_SIZE is exact.
Usage: python tools/traffic_post.py gpurun_out/traffic/traffic_raw.json profiles/<tag>_scan_hbm_traffic.json"""
import json
import sys

raw = json.load(open(sys.argv[1]))
out = {'_method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over tools/traffic_probe.py '
                  '(tools/prof_traffic.sh); units KB*1024; 512 MB flush before each kernel; FETCH_SIZE x2 (gfx950 '
                  'streaming-read under-count, see the calib_copy_* entries of the same run); WRITE_SIZE exact.'}
for name, e in raw.items():
    f = sum(sum(k['FETCH_SIZE_KB'][-1:]) for k in e['kernels'].values()) * 1024.0
    w = sum(sum(k['WRITE_SIZE_KB'][-1:]) for k in e['kernels'].values()) * 1024.0
    alg = e['algorithmic_read'] + e['algorithmic_write']
    out[name] = {'algorithmic_bytes': alg, 'algorithmic_read': e['algorithmic_read'], 'algorithmic_write': e['algorithmic_write'],
                 'FETCH_SIZE_bytes_raw': f, 'WRITE_SIZE_bytes': w, 'hbm_read_bytes_corrected_x2': 2 * f,
                 'hbm_traffic_bytes': 2 * f + w, 'traffic_over_algorithmic': (2 * f + w) / alg,
                 'kernels': [k for k in e['kernels']]}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for k, v in out.items():
    if k != '_method':
        print(k, 'traffic/algorithmic = %.4f' % v['traffic_over_algorithmic'])
