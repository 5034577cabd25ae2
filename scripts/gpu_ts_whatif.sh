cd $GRAFT_REPO_ROOT
for m in 2048 2049 2080 3072 2056 6144 10240 14336; do echo "=== mask $m"; TS_LIB=libtcr_w$m.so TS_KNOBS=0 python scripts/fused_ts.py 2>&1 | grep -v amdgpu; done
