O=gpurun_out/r5i; mkdir -p $O
for v in 0 1; do echo "SSRHIP_GEMVM_XFIRST=$v" | tee -a $O/gemvm_lab_fine.log; SSRHIP_GEMVM_XFIRST=$v timeout 120 tools/bin/gemvm_lab 16 2>&1 | tee -a $O/gemvm_lab_fine.log; done
