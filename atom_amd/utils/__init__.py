from .kvcache import BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4  # noqa: F401
