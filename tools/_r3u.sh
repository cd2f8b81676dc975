tools/probes/bin/tail_probe 20 3 0
tools/probes/bin/tail_probe 20 3 0
tools/probes/bin/tail_probe 20 3 1
SBBSEG_TAIL_X3_PS=0 tools/probes/bin/tail_probe 20 3 0
