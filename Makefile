# convenience targets; the driver uses __graft_entry__.build(), pytest and bench.py directly
.PHONY: build test test-gpu bench clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -x -q -m "not gpu"
test-gpu: build
	python -m pytest tests -x -q -m gpu
bench: build
	python bench.py
clean:
	rm -f svinet_amd/lib/*.so svinet_amd/bin/svinet oracle/*.so
