mkdir -p gpurun_out/r5j
O=gpurun_out/r5j
{
echo "== WENO 256^3, 1 source: library (chunks of 8) / variant with chunks of 16 everywhere"
python scripts/weno_time.py 256 | tail -1
TTCR_AMD_LIB=$PWD/variants/c16.so python scripts/weno_time.py 256 | tail -1
echo "== WENO 256^3, 8 sources"
python scripts/weno_batch.py 256 8
TTCR_AMD_LIB=$PWD/variants/c16.so python scripts/weno_batch.py 256 8
echo "== occupancy cap (dynamic LDS that keeps two workgroups per CU) below N batch entries: default 2 (lone source only)"
for b in 0 2 3 5; do for n in 1 2 4; do TTCR_FSM_XS_LDS_BELOW=$b python scripts/lone_time.py 512 2 $n | sed "s/^/cap below $b: /"; done; done
for b in 0 2 5; do for n in 1 4; do TTCR_FSM_XS_LDS_BELOW=$b python scripts/lone_time.py 256 3 $n | sed "s/^/cap below $b: /"; done; done
} > $O/weno_c16_cap.txt 2>&1
sed 's/ lib=lib[a-z0-9_.]*//; s/ pair=default//' $O/weno_c16_cap.txt
{
echo "== source pairs on chunks of 16 levels with two workgroups per CU (variant: -DFSM_CHUNK3=16 -DFSM_PAIR16_MINW=2) against the library"
python scripts/lone_time.py 512 2 8
TTCR_AMD_LIB=$PWD/variants/c16p.so python scripts/lone_time.py 512 2 8
python scripts/lone_time.py 512 2 16
TTCR_AMD_LIB=$PWD/variants/c16p.so python scripts/lone_time.py 512 2 16
ITERS=1,2 python scripts/lone_skip.py 512 64
TTCR_AMD_LIB=$PWD/variants/c16p.so ITERS=1,2 python scripts/lone_skip.py 512 64
} > $O/pairs_c16.txt 2>&1
sed 's/ lib=lib[a-z0-9_.]*//; s/ pair=default//' $O/pairs_c16.txt
