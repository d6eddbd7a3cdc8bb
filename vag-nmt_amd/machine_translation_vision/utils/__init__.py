"""utils of the reference (utils/__init__.py: ``__all__ = ['utils', 'im_retrieval_eval']``)."""
__all__ = ['utils', 'im_retrieval_eval']
