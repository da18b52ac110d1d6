"""Model registry with the reference's surface: ``from models import get_model`` (reference
``models/__init__.py:11-24``).  Only the model families on the accelerated hot path are registered
(SURVEY.md section 8a, 8f): PathCNN and the stale DeeperGCN copies ('multiomix') are out of scope for this
library."""
from .deepergcn import DeeperGCN
from .multilevel_gnn import MultilevelGNN
from .multilevel_gnn_seq import MultilevelGNNSeq, PathwayHeadSeq  # noqa: F401
from .vae import VAE, VQ_VAE, AutoEncoder, VectorQuantizer  # noqa: F401
from .diff_pooling import DiffPool, DiffPoolLayer, SAGEConvolutions  # noqa: F401

MODELS = {
    'deepergcn': DeeperGCN,
    'multilevel_gnn': MultilevelGNN,
    'multilevel_gnn_seq': MultilevelGNNSeq,
    'vae': VAE,
    'mmd_vae': VAE,
    'vq_vae': VQ_VAE,
    'autoencoder': AutoEncoder,
}


def get_model(model_name):
    return MODELS[model_name]
