#!/bin/bash
cd /root/repo
for b in wgrad_body_probe wgp_noload; do for n in 4 16 32; do
echo "== $b nrdb $n"; timeout 60 tools/$b $n 2 0 1 0 | grep "timeline\|loop "
done; done
