"""Native-kernel Llama building blocks with the call graph of the reference's e2e tree
(e2e/punica-atom/punica/models/llama.py): the 4-tuple (outlier, norms, outlier_scales, norm_scales) flows between ops."""
from .llama import (LinearInt4, LlamaAttention, LlamaDecoderLayer, LlamaForCausalLM, LlamaMLP, LlamaModel,  # noqa: F401
                    LlamaRMSNorm, LlamaRMSNormInt4)
