for k in 0 2 4 8 0; do echo "stagger $k"; tools/probes/bin/block_probe_s$k; done
