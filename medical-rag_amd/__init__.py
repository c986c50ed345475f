"""MI355X-native local dense retriever for MedTutor (yale-nlp/medical-rag).

The directory name carries a hyphen; import it as `medical_rag_amd` through the
shim module of that name at the repository root.
"""
from . import _native, local_rag  # noqa: F401
from .local_rag import (PageIndex, configure, embed_queries, get_surrounding_pages, load_embeddings,  # noqa: F401
                        retrieve_top_index, search)

__version__ = "0.1.0"
