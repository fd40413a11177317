#!/bin/bash
# compact kernel resource usage: tools/kres.sh file.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I/root/repo -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
 | grep -E "error|Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|SGPRs:|LDS Size" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - - - \
 | sed -E 's/Function Name: _ZN3sfa12_GLOBAL__N_1[0-9]+//; s/EvNS_[0-9A-Za-z]+E//' 
