# stream-count / priority experiment: bash tools/exp/streams.sh (on the GPU box)
# measured (MI355X, ms per step): 4 streams 32.9-33.1 at either priority; a 5th stream (S4F_EAGER_STREAM=new) 32.7-32.9 with
# equal priorities but 40.4 with the high-priority chain (S4F_MAIN_PRIORITY=1); GPU_MAX_HW_QUEUES=8 does not change that.
run() { echo "$1: $(env $2 timeout -k 10 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"; }
for i in 1 2; do
run "4 streams, equal priorities (eager SGD on the side stream)" "S4F_X=1"
run "4 streams, high-priority chain" "S4F_MAIN_PRIORITY=1"
run "5 streams, equal priorities (eager SGD on a new stream)" "S4F_EAGER_STREAM=new"
run "5 streams, high-priority chain" "S4F_EAGER_STREAM=new S4F_MAIN_PRIORITY=1"
run "4 streams, equal priorities, no eager SGD" "S4F_EAGER_SGD=0"
done
