#!/bin/bash
# Development aid: timeline of k_post_chain_r's stages from builds that each carry ONE extra stamp (more stamps at once put
# the kernel's registers into scratch memory and change what is measured); raw stamps in gpurun_out/stamps_<mask>.csv
mkdir -p gpurun_out; : > gpurun_out/timeline.txt
for m in 0x603 0x609 0x611 0x641 0x681; do
  GSMCAL_DEVTIMING_DUMP=gpurun_out/stamps_$m.csv python tools/devtiming.py --light 2 --mask $m 2>&1 | grep -A1 "post_chain" >> gpurun_out/timeline.txt
done
cat gpurun_out/timeline.txt | cut -c1-400
