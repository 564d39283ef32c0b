set -e
timeout -k 10 120 python scripts/fwd_fullres.py 1 bf16
ERFNET_NO_PAIR=1 timeout -k 10 120 python scripts/fwd_fullres.py 1 bf16
timeout -k 10 120 python scripts/fwd_fullres.py 4 bf16
ERFNET_NO_PAIR=1 timeout -k 10 120 python scripts/fwd_fullres.py 4 bf16
