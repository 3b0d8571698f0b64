# usage: tools/griddiv.sh  -- sweep workgroups-per-launch divisor x frames in flight x hardware queues
for q in 24 32 64; do for p in 3; do for od in "16 5" "16 6" "16 8" "16 10" "16 12"; do set -- $od
echo "queues $q persist $p overlap $1 griddiv $2: $(GPU_MAX_HW_QUEUES=$q TPT_GRID_DIV=$2 timeout 100 python bench.py --no-cpu-baseline --persistent $p --overlap $1 2>&1 | tail -1 | cut -c1-60)"; done; done; done
for od in "16 6" "16 8" "16 10"; do set -- $od
echo "queues 32 persist 1 overlap $1 griddiv $2: $(GPU_MAX_HW_QUEUES=32 TPT_GRID_DIV=$2 timeout 100 python bench.py --no-cpu-baseline --persistent 1 --overlap $1 2>&1 | tail -1 | cut -c1-60)"; done
