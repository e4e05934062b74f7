"""CPU oracle for the TSM-R50 clip-inference hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``workoutdetector_amd/`` may import this package: the product path
is the HIP engine behind ``include/tsm_hip.h`` and it must fail loudly when the
extension is missing.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use the oracle, and only as the checker /
the timed CPU baseline.

Parity status (see DESIGN.md "Oracle"):
  * ``counting_oracle.pred_to_count``, ``tsm_oracle.temporal_shift`` and the
    segment-consensus head are PINNED: against the reference's own known-answer
    vectors (tests/test_inference_count.py:8-48, the docstring example
    utils/inference_count.py:141-143, notebooks/rep_analysis.ipynb cell 18) and
    against outputs of the reference's own function bodies executed in the build
    container (tests/golden/make_reference_vectors.py -> tests/golden/ref_*.json).
  * ResNet-50 arithmetic lives in torchvision 0.13.0 / onnxruntime (absent from
    /root/reference and from the image): it is restated from the public
    definition.  The reference holds no numeric logits fixture, so *logits*
    parity is "unpinned by the reference"; golden logits under tests/golden/ are
    this oracle's outputs on seeded inputs.  The trunk wiring is cross-checked
    against an independent public ResNet-50 v1.5 (HuggingFace transformers,
    tests/test_oracle_wiring.py).
"""
