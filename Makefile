# Convenience targets; the driver uses __graft_entry__.build() / pytest / bench.py directly.
.PHONY: build test test-gpu bench clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -x -q -m "not gpu"
test-gpu: build        # on an MI355X box
	python -m pytest tests -x -q -m gpu
bench: build           # on an MI355X box; one JSON line on stdout
	python bench.py
clean:
	$(MAKE) -C formula-vad_amd/csrc clean
	$(MAKE) -C oracle clean
	$(MAKE) -C tools clean
