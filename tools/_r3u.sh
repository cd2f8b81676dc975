for k in 0 64 0 64; do timeout 120 tools/probes/bin/block_probe_$k; done
