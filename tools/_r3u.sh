tools/probes/bin/tail_probe 140 3 0
tools/probes/bin/tail_probe 140 3 0
