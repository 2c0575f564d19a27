# development (through gpurun): the headline solve launch by progress of its agents -- iteration cap 0 = set-up + crash-start batch + one pivot scan + outputs
b() { echo "$*: $(python bench.py --no-cpu-baseline --no-secondary --steps 18 --warmup 9 $(for o in "$@"; do echo --debug-option $o; done) 2>/dev/null | tail -1 | python tools/bench_brief.py)"; }
b iter_cap=0 crash_min=99
b iter_cap=0
b iter_cap=1 crash_min=99
b iter_cap=2 crash_min=99
b iter_cap=4 crash_min=99
b iter_cap=0 no_persist=1
b iter_cap=0 no_split_t=1
b iter_cap=2000 no_split_t=1
b iter_cap=2000 no_persist=1
