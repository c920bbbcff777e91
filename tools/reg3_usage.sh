#!/bin/bash
# VGPR / spill report of the register3_kernel instantiations (compile only; no GPU needed)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function $EXTRA -S --cuda-device-only /root/repo/tbv_slam_public_amd/csrc/register.hip -o /tmp/reg.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning:|register3" -A12 | grep -E "error|warning:|Function Name|VGPRs:|Spill|ScratchSize" | sed 's/.*remark: //; s/ \[-Rpass.*//' | tr '\n' ' ' | sed 's/Function Name/\nFunction Name/g'; echo
