cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q -k "fp32_class" 2>&1 | tail -5 | cut -c1-300
