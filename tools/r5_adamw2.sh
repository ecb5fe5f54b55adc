# the update kernel from process to process on one box (see csrc/optim.hip: adamw_table_kernel)
for i in 1 2 3; do
python3 bench.py --no-cpu-baseline --steps 24 2>/dev/null | tail -n 1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); a=[k for k in d['kernel_table'] if 'adamw' in k['kernel']][0]; print('RES', d['ms_per_step'], d['box']['d2d_copy_gbs'], a['kernel'][:30], a['avg_launch_us'], a['achieved'])"
done
