"""textreact_amd -- MI355X-native retrieval hot path of TextReact (exact flat k-NN on gfx950).

Only the hot path lives here: the HIP kernels + C ABI (csrc/, include/trx_knn.h), the FAISS-shaped
host objects the reference's retrieval script binds (faiss_compat), the retrieval CLI mirror
(retrieve_faiss) and the row-sharded multi-GPU search (sharded).
"""
__version__ = "0.1.0"
