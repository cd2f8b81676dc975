tools/probes/bin/tail_probe 20 3 0
tools/probes/bin/tail_probe 20 3 0
tools/probes/bin/tail_probe 20 2 0
tools/probes/bin/tail_probe 20 2 0
