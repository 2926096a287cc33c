for sp in 0 16 32 64; do echo "== DVIS_CU_SPLIT=$sp"; DVIS_CU_SPLIT=$sp timeout 600 python tools/rank_emulation.py --worlds 1,8 --batches 1,2 2>&1 | grep -v amdgpu.ids | tail -5; done
