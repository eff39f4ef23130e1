cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep "Counter_Name" | grep -i "SPI_\|SQ_BUSY_CU\|SQ_WAVES\|SQ_LEVEL_WAVES\|GRBM_SPI\|SQ_WAVES_" | head -80
rocm-smi --showpids 2>/dev/null | head -20
rocminfo 2>/dev/null | grep -i "compute unit\|Max Waves\|Marketing\|Shader Arrs\|SIMDs per" | head -12
