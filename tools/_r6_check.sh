cd /root/repo
python tools/kbench.py --map-scans 20 --reps 10 --check 2>&1 | grep -E "rtcsm|C=|check|rror"
python tools/kbench.py --map-scans 20 --reps 10 2>&1 | grep -E "rtcsm |rror"
python tools/c5bench.py --reps 3 2>&1 | grep -E "config5|check|rror"
python -m pytest tests/test_gpu_full_size.py -x -q 2>&1 | tail -3
