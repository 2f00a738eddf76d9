"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain torch ops) of the reference's per-frame streaming-inference path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import anything from here, and only as the checker / the reported CPU baseline -- never as
the product path.  The product (``aha-_amd/``) must not import this package.

Pinning: the reference has no tests or golden vectors for this path and its
model code is not importable here (absent llava/peft/deepspeed/wandb), so the restatement
is pinned by (1) the local transformers Qwen2Model / SiglipVisionModel on seeded configs,
(2) the reference's own cache classes test/{sink,sliding_window,static}_cache.py imported
from /root/reference, and (3) the golden fixtures under tests/golden/ generated from (1)
and (2) by tests/make_golden.py.  See DESIGN.md "Oracle".
"""
