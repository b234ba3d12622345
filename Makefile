# repo-level conveniences (the product builds with `make -C zkevm-circuits_amd`, the oracle with `make -C oracle`)
PYTHON ?= python
GPU ?=

all:
	$(MAKE) -C zkevm-circuits_amd -s
	$(MAKE) -C oracle -s

# T1 kit (shim/t1_standalone/README.md): kit inputs + self-check proofs, verified from the files alone, packed with the Rust
# program that puts them in front of upstream verify_proof.  `make t1-kit GPU=--gpu` proves through libzkmi355.so.
t1-kit:
	$(MAKE) -C oracle -s
	rm -rf t1_kit t1_kit.tar.gz
	$(PYTHON) tools/t1_kit.py make t1_kit $(GPU)
	$(PYTHON) tools/t1_kit.py check t1_kit
	tar czf t1_kit.tar.gz t1_kit shim/t1_standalone tools/t1_kit.py tests/plonk_fixtures.py
	@echo "t1_kit.tar.gz ready: follow shim/t1_standalone/README.md (4 commands)"

.PHONY: all t1-kit
