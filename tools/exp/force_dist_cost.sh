#!/bin/bash
# r4: cost of the collective step with ONE rank (the all-reduce moves nothing): fused single-rank step vs --force-dist through
# torch.distributed's stream (default since round 5) vs our own communicator on the compute stream (--direct-rccl, opt-in).
for i in 1 2; do
python bench.py --steps 100 --warmup 20 --no-traffic --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused single rank   ', d['ms_per_step'])"
python bench.py --steps 100 --warmup 20 --no-traffic --no-secondary --no-cpu-baseline --force-dist 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-dist torch    ', d['ms_per_step'], d['config']['collective'][-40:])"
python bench.py --steps 100 --warmup 20 --no-traffic --no-secondary --no-cpu-baseline --force-dist --direct-rccl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-dist direct   ', d['ms_per_step'], d['config']['collective'][-40:])"
done
