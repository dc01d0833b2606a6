#!/bin/bash
# what kind of box this is, UNDER LOAD: clocks sampled while a bench runs, the driver's VM parameters, VRAM use, uptime
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; T=${1:-box}; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
{
echo "# uptime: $(cat /proc/uptime)"; uname -r
for p in vm_fragment_size vm_block_size vm_size vm_update_mode noretry mtype_local; do echo "amdgpu.$p = $(cat /sys/module/amdgpu/parameters/$p 2>/dev/null)"; done
for f in /sys/class/drm/card*/device/mem_info_vram_used /sys/class/drm/card*/device/mem_info_vram_total /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition; do echo "$f = $(cat $f 2>/dev/null)"; done
cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
( timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --steps 20000 > $O/${T}_bench.json 2> $O/${T}_bench.err ) &
BP=$!
sleep 4
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks --showpower 2>&1 | grep -i "fclk\|mclk\|sclk\|Power (W)" | tr '\n' ' '; echo; sleep 1.5; done
wait $BP
python - $T <<'PY'
import json,sys
j=json.loads(open('gpurun_out/%s_bench.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print('bench', round(j['ms_per_step']*1000,1), 'copy', round(j['roofline']['measured_copy_GBs']), k, j.get('effective_sclk_mhz'))
PY
} > $O/${T}_box.txt 2>&1
cat $O/${T}_box.txt
