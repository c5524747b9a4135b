"""Side-by-side table of scripts/gpu_conv_ab.sh's timing.txt (one conv_bench block per library)."""
import re, sys
t = open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/conv_ab/timing.txt').read()
res, cur = {}, []
for line in t.splitlines():
    m = re.match(r'\s+(.*?)\s+([\d.]+) ms\s+([\d.]+) TF/s\s+([\d.]+) of roof\s+bits (\w+)', line)
    if m:
        cur.append((m.group(1), float(m.group(2)), float(m.group(4)), m.group(5)))
    m = re.match(r'(\S+)\s+frames (\d+): ([\d.]+) TF/s.*\(([\d.]+) of roof', line)
    if m:
        res[m.group(1).split('/')[-1]] = (cur, m.group(4)); cur = []
libs = list(res)
print(' ' * 36, *[f"{l[:15]:>15s}" for l in libs])
for i in range(len(res[libs[0]][0])):
    print(f"{res[libs[0]][0][i][0]:36s}", *[f"{res[l][0][i][1]:8.3f}  {res[l][0][i][2]:.3f}" for l in libs],
          'same bits' if len({res[l][0][i][3] for l in libs}) == 1 else 'DIFFERENT BITS')
print(f"{'listed layers, of roof':36s}", *[f"{res[l][1]:>15s}" for l in libs])
