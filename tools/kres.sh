#!/bin/bash
# registers / scratch / occupancy of the kernels whose name matches $1 (compiles the library once into /tmp)
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I/root/repo/include -I/root/repo/ropebwt2_amd/csrc -Wno-unused-value -Rpass-analysis=kernel-resource-usage -o /tmp/kres.so /root/repo/ropebwt2_amd/csrc/rb2_engine.hip -ldl -lpthread 2> /tmp/kres.txt
grep -A12 "Function Name: .*$1" /tmp/kres.txt | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|VGPRs Spill|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//'
