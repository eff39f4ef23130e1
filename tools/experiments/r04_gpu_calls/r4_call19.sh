cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|ifetch\|SQC_\|INST_CACHE\|SQ_INSTS_\|SQ_WAIT_INST\|SQ_WAVE_" | head -80
