#!/bin/bash
# Does co-residency hide the fused epilogues?  FFN-1 / d_FFN-2 shapes of the B = 64 step with the plain, bias, gelu' and mul-aux
# epilogues on (a) the default dispatch (256-square two-phase tile, one workgroup per CU), (b) the 128-square tile (2-3 workgroups
# per CU), (c) the 128-square K-group tile off.  Kernel time under hipGraph replay (tools/gemm_bench.py).
out=gpurun_out/r4g; mkdir -p $out
S="nt:5120x3072x768:none:bf16 nt:5120x3072x768:bias:bf16 nt:5120x3072x768:gelugrad:bf16 nt:11520x3072x768:none:bf16 nt:11520x3072x768:gelugrad:bf16 nt:2752x3072x768:none:bf16 nt:2752x3072x768:gelugrad:bf16 nn:5120x3072x768:none:bf16 nn:5120x3072x768:mulaux:bf16 nn:11520x3072x768:mulaux:bf16"
{
echo "## default dispatch"; GRAPH=1 timeout 300 python tools/gemm_bench.py $S
echo "## HAMT_P8=0 HAMT_FAST_BM=128 (128 x 128 tiles, 4 waves, 2-3 workgroups per CU)"; GRAPH=1 HAMT_P8=0 HAMT_FAST_BM=128 HAMT_KG=1 timeout 300 python tools/gemm_bench.py $S
echo "## HAMT_P8=0 HAMT_FAST_BM=256 (one-phase 256-square tile)"; GRAPH=1 HAMT_P8=0 HAMT_FAST_BM=256 HAMT_KG=1 timeout 300 python tools/gemm_bench.py $S
} > $out/epi_overlap.txt 2>&1
tail -50 $out/epi_overlap.txt
