export PYTHONPATH=$PWD
for e in 0 1; do echo "== DTF_F32_EMULATION=$e"; DTF_F32_EMULATION=$e timeout -k 10 120 python scripts/dbg/fc_bwd_roles_time.py 2>&1 | grep -v amdgpu.ids | head -7; done
