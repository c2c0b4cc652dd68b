"""graphvqa_amd -- MI355X-native scene-graph execution path for GraphVQA.

Only what the hot path needs: `csrc/` (HIP kernels + C ABI, built into lib/libgvqa_hip.so),
the ctypes binding, the graph container and host-side mirrors of the reference's operator
interface (`gat_skip`, ...).  Importing the package does not load the HIP library; constructing
an operator does, and fails loudly when it is missing.
"""
# one module per reference interface on (or next to) the path; tests/test_host.py holds this list to the directory
__all__ = ["gat_skip",          # gat, gat_seq                      (reference gat_skip.py)
           "lcgn",              # lcgn_seq                          (baseline_and_test_models/lcgn.py)
           "baseline_models",   # gine_seq, gcn_seq                 (pipeline_model_gine.py / _gcn.py, inline classes)
           "sg_encoder",        # GroundTruth_SceneGraph_Encoder    (pipeline_model_gat.py:553-610)
           "pipeline_head",     # MyConditionalGlobalAttention, ShortAnswerClassifier (pipeline_model_gat.py:108-185,722-728)
           "scene_graph",       # scene-graph JSON -> batch         (gqa_dataset_entry.py:190-372,631-675)
           "graph",             # SceneGraphBatch, HostLayout: the CSR handle
           "parallel",          # graph sharding + the one all-gather
           "synth",             # seeded synthetic batches / parameters (bench + tests)
           "build", "_lib"]     # hipcc build of lib/libgvqa_hip.so; its ctypes binding
