# Convenience targets (the driver uses __graft_entry__.build / pytest / bench.py directly).
PY ?= python

.PHONY: build test test-gpu bench smoke clean golden

build:            ## libagx.so (hipcc, gfx950), libagx_runner.so (g++), the C oracle
	$(PY) -c "import __graft_entry__ as g; g.build()"

test:             ## CPU suite: oracle vs goldens, host logic, ABI surface, 2-rank gloo sharding
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu:         ## parity through the C ABI on an MI355X
	$(PY) -m pytest tests -x -q -m gpu

smoke:
	$(PY) -c "import __graft_entry__ as g; g.build(); g.smoke()"

bench:            ## env steps/s + roofline + cpu_baseline JSON line
	$(PY) bench.py

golden:           ## regenerate tests/golden from the reference checkout (build container only)
	$(PY) tests/golden/make_golden.py

clean:
	rm -rf active-gym_amd/lib oracle/_build tools/membench .pytest_cache
