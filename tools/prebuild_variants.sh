#!/bin/bash
# Build the product library and every variant the GPU suite loads (so that they travel with the gpurun snapshot instead of being
# compiled on the GPU box): hooks (-DMFAS_TEST_HOOKS), adamlib, dma, asan.  Variants are rebuilt only when the source digest changed.
cd "$(dirname "$0")/.." || exit 1
python - <<'PY'
import __graft_entry__ as g
from concurrent.futures import ThreadPoolExecutor
g.build()
jobs = [lambda: g.build_variant("hooks", ["-DMFAS_TEST_HOOKS"]), lambda: g.build_variant("adamlib", ["-DMFAS_ADAM_LIBRARY_FORMS"]),
        lambda: g.build_variant("dma", ["-DMFAS_RES_DMA=1"]), g.build_asan]
with ThreadPoolExecutor(4) as ex:
    for r in ex.map(lambda f: f(), jobs):
        print("built", r)
PY
