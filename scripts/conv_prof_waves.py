"""Per-wave phases of the persistent patch kernels from scripts/micro/conv_prof (variant library = profiles/r06_conv_prof.patch
on lwm_amd/csrc): python scripts/conv_prof_waves.py gpurun_out/conv_prof/*.bin"""
import sys
import numpy as np
for path in sys.argv[1:]:
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 8, 8).astype(np.int64)
    a = a[a[:, 0, 0] > 0]
    print(path, len(a), 'tiles')
    for n, i in (('patch landed (tile begins -> barrier)', 0), ('main loop', 1), ('sums + barrier + next patch requested', 2), ('stores issued', 3)):
        v = a[:, :, i + 1] - a[:, :, i]
        print(f'   {n:40s} per wave index:', np.round(v.mean(0)).astype(int))
    end = a[:, :, 2]
    print('   main loop ends, last wave minus first: mean %d cycles; slowest wave index histogram %s' % ((end.max(1) - end.min(1)).mean(), np.bincount(end.argmax(1), minlength=8)))
    key = a[:, 0, 7] * 4096 + (a[:, 0, 5] & 0xfff)
    per = [(a[key == k][:, 0, 4].max() - a[key == k][:, 0, 0].min()) / (key == k).sum() for k in np.unique(key)]
    print('   cycles per tile and workgroup: %.0f  (MFMA cycles of a SIMD per tile: 294912)' % np.mean(per))
