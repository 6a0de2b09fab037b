from .cat_tensor import BatchLenInfo  # noqa: F401
from .kvcache import BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4  # noqa: F401
