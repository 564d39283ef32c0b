set -e
timeout -k 10 60 python scripts/micro_pair.py --c 128 --h 64 --w 128 --d 2 --reps 50 --axis 0
timeout -k 10 60 python scripts/micro_pair.py --c 64 --h 128 --w 256 --d 1 --reps 50 --axis 0
timeout -k 10 60 python scripts/micro_pair.py --c 128 --h 64 --w 128 --d 2 --reps 50 --axis 1
