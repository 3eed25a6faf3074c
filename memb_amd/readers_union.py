"""Several readers behind one reader's interface.

Interface and results of the reference's `memb.ReadersUnion`
(python/memb/readers_union.py:41-104): mode 'concatenate' lays the readers'
vectors side by side, mode 'average' takes their float32 mean; a word that one
reader does not know contributes zeros there. Differences are only in how the
rows get where they belong: for concatenation every reader decodes straight into
its column block of the result (the C ABI's `ld` / `col_off`), and
`batch_embedding_device` does either merge without leaving the GPU.
"""
import numpy as np

from .reader import BaseReader, Reader, tokenizer_word_list

CONCATENATE = 'concatenate'
AVERAGE = 'average'
MODES = (AVERAGE, CONCATENATE)


def _merge_host(mode, pieces):
    """numpy merge of per-reader results (1-D vectors or 2-D batches)"""
    if mode == CONCATENATE:
        return np.concatenate(pieces, axis=-1)
    return np.mean(pieces, axis=0)


class ReadersUnion(BaseReader):
    """readers : at least two readers (for 'average' all of one dimension)
    mode    : 'average' or 'concatenate'
    `dim` is the dimension of the merged vectors."""

    def __init__(self, readers, mode):
        super().__init__()
        if len(readers) < 2:
            raise AssertionError('You must pass at least 2 readers to create a union')
        if mode not in MODES:
            raise KeyError('Mode {} is not supported. Available modes are {}'.format(mode, list(MODES)))
        widths = [reader.dim for reader in readers]
        if mode == AVERAGE and len(set(widths)) != 1:
            raise AssertionError('Dimensions of all readers must be equal for average mode')
        self._readers = list(readers)
        self._mode = mode
        self._widths = widths
        self._word_batch = None   # packed query words of batch_embedding_device, shared by the readers

    @property
    def dim(self):
        return sum(self._widths) if self._mode == CONCATENATE else self._widths[0]

    def keys(self):
        """Every word at least one reader knows, sorted"""
        merged = set()
        for reader in self._readers:
            merged.update(reader.keys())
        return sorted(merged)

    def word_embedding(self, word):
        return _merge_host(self._mode, [reader.word_embedding(word) for reader in self._readers])

    def batch_embedding(self, words):
        native = all(isinstance(reader, Reader) for reader in self._readers)
        if self._mode == CONCATENATE and native:
            # no concatenation pass: each reader fills its own columns of the result
            merged = np.empty((len(words), self.dim), dtype=np.float32)
            column = 0
            for reader, width in zip(self._readers, self._widths):
                reader.batch_embedding_into(words, merged, column)
                column += width
            return merged
        return _merge_host(self._mode, [reader.batch_embedding(words) for reader in self._readers])

    def batch_embedding_device(self, words):
        """batch_embedding merged on the GPU, returned as a torch.Tensor (not in the
        reference API). Concatenation: column blocks of one (n, dim) tensor. Average:
        reader 1 stores, readers 2..R add, the last one also divides by R -- the very
        additions and the one division numpy.mean performs, in its order, so the
        result has the same bits as batch_embedding. All readers on one device."""
        import torch
        readers = self._readers
        if not all(isinstance(reader, Reader) for reader in readers):
            raise TypeError('device merge needs memb_amd.Reader instances')
        if len({reader.device for reader in readers}) != 1:
            raise ValueError('all readers of a union must be on one device')
        from . import _memb
        device = 'cuda:{}'.format(readers[0].device)
        stream = torch.cuda.current_stream(torch.device(device)).cuda_stream
        # the words are packed and copied ONCE and resolved on the GPU by every reader's own hash table
        # (Reader.resolve_rows_device): the row ids never visit the host
        row_ids = [torch.empty((len(words),), dtype=torch.int32, device=device) for _ in readers]
        if self._word_batch is None:
            self._word_batch = _memb.WordBatch(readers[0].device)
        _memb.union_words_to_rows_device(
            self._word_batch, words, [reader._impl for reader in readers], [ids.data_ptr() for ids in row_ids], stream)
        merged = torch.empty((len(words), self.dim), dtype=torch.float32, device=device)
        # one launch that decodes every reader's words of a tile and writes the merged rows once,
        # where the readers can share a kernel
        concatenate = self._mode == CONCATENATE
        columns = [sum(self._widths[:i]) if concatenate else 0 for i in range(len(readers))]
        if len(words) and _memb.union_rows_to_device(
                [reader._impl for reader in readers], [ids.data_ptr() for ids in row_ids], columns,
                len(words), merged.data_ptr(), merged.stride(0), stream, not concatenate):
            return merged
        if concatenate:
            column = 0
            for reader, ids, width in zip(readers, row_ids, self._widths):
                reader.rows_embedding_device(ids, out=merged, col_off=column)
                column += width
            return merged
        count = len(readers)
        for position, (reader, ids) in enumerate(zip(readers, row_ids)):
            reader.rows_embedding_device(
                ids, out=merged, accumulate=position > 0, divisor=float(count) if position == count - 1 else 0.0)
        return merged

    def tokenizer_embedding(self, tokenizer):
        """Embedding-layer weights for a keras Tokenizer, merged over the readers"""
        return self.batch_embedding(tokenizer_word_list(tokenizer))

    def tokenizer_embedding_device(self, tokenizer):
        """tokenizer_embedding merged on the GPU (see batch_embedding_device)"""
        return self.batch_embedding_device(tokenizer_word_list(tokenizer))
