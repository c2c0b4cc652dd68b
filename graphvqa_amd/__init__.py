"""graphvqa_amd -- MI355X-native scene-graph execution path for GraphVQA.

Only what the hot path needs: `csrc/` (HIP kernels + C ABI, built into lib/libgvqa_hip.so),
the ctypes binding, the graph container and host-side mirrors of the reference's operator
interface (`gat_skip`, ...).  Importing the package does not load the HIP library; constructing
an operator does, and fails loudly when it is missing.
"""
__all__ = ["synth", "scene_graph", "gat_skip", "graph", "build"]
