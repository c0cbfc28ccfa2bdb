O=gpurun_out/r4a; mkdir -p $O
python tools/fused_prof.py 314 2>&1 | grep -v amdgpu.ids | tee $O/fused_prof_314.log
python tools/fused_prof.py 700 2>&1 | grep -v amdgpu.ids | tee $O/fused_prof_700.log
