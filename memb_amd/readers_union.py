import numpy as np

from .reader import BaseReader, Reader, tokenizer_word_list


class AverageUnionMaker:
    # reference python/memb/readers_union.py:5-18
    @staticmethod
    def check(readers):
        dims = [reader.dim for reader in readers]
        if any(dim != dims[0] for dim in dims):
            raise AssertionError('Dimensions of all readers must be equal for average mode')

    @staticmethod
    def dim(reader_dims):
        return reader_dims[0]

    @staticmethod
    def merge(vectors):
        return np.mean(vectors, axis=0)

    @staticmethod
    def batch(readers, words):
        return np.mean([reader.batch_embedding(words) for reader in readers], axis=0)


class ConcatenatedUnionMaker:
    # reference python/memb/readers_union.py:21-32
    @staticmethod
    def check(readers):
        pass

    @staticmethod
    def dim(reader_dims):
        return sum(reader_dims)

    @staticmethod
    def merge(vectors):
        return np.concatenate(vectors, axis=-1)

    @staticmethod
    def batch(readers, words):
        # Every reader decodes straight into its own column block of the merged
        # matrix (leading dimension = sum of dims), so no concatenation pass.
        if not all(isinstance(reader, Reader) for reader in readers):
            return np.concatenate([reader.batch_embedding(words) for reader in readers], axis=-1)
        dims = [reader.dim for reader in readers]
        out = np.empty((len(words), sum(dims)), dtype=np.float32)
        col_off = 0
        for reader, dim in zip(readers, dims):
            reader.batch_embedding_into(words, out, col_off)
            col_off += dim
        return out


UNION_MAKERS = {
    'average': AverageUnionMaker(),
    'concatenate': ConcatenatedUnionMaker(),
}


class ReadersUnion(BaseReader):
    '''ReadersUnion is a wrapper that makes a list of Readers behave just like
    one. It returns either average or concatenation of embeddings obtaied from
    the readers it contains (reference python/memb/readers_union.py:41-104).
    Parameters
    ----------
    readers : list of Reader

    mode : str
        Strategy to use for merging vectors. Can be either 'average' or 'concatenate'
    Attributes
    ----------
    dim : int
        Dimension of vectors after merge
    '''
    def __init__(self, readers, mode):
        super().__init__()

        if len(readers) < 2:
            raise AssertionError('You must pass at least 2 readers to create a union')

        self._union_maker = UNION_MAKERS.get(mode)
        if self._union_maker is None:
            raise KeyError('Mode {} is not supported. Available modes are {}'.format(
                mode, list(UNION_MAKERS.keys())))

        self._union_maker.check(readers)
        self._readers = readers

    @property
    def dim(self):
        return self._union_maker.dim([reader.dim for reader in self._readers])

    def keys(self):
        '''Union of keys contained in wrapped models'''
        all_keys = set()
        for reader in self._readers:
            all_keys |= set(reader.keys())

        return sorted(all_keys)

    def word_embedding(self, word):
        '''Merged vectors from all readers for a single word
        Parameters
        ----------
        word : str
        '''
        return self._union_maker.merge([reader.word_embedding(word) for reader in self._readers])

    def batch_embedding(self, words):
        '''Merged vectors from all readers for a list of words
        Parameters
        ----------
        words : list of str
        '''
        return self._union_maker.batch(self._readers, words)

    def batch_embedding_device(self, words):
        '''batch_embedding merged on the GPU; returns a torch.Tensor (DLPack capable).
        concatenate: every reader decodes into its column block of one (n, sum of dims)
        tensor; average: readers 2..R add to the first one's rows, the last one divides
        by R -- the additions and the division numpy.mean performs, in the same order.
        All readers must sit on the same device (not in the reference API).'''
        import torch
        readers = self._readers
        if not all(isinstance(reader, Reader) for reader in readers):
            raise TypeError('device merge needs memb_amd.Reader instances')
        device = 'cuda:{}'.format(readers[0].device)
        if any(reader.device != readers[0].device for reader in readers):
            raise ValueError('all readers of a union must be on one device')
        rows = [torch.from_numpy(reader.resolve_rows(words).view('int32')).to(device) for reader in readers]
        out = torch.empty((len(words), self.dim), dtype=torch.float32, device=device)
        if self._union_maker is UNION_MAKERS['concatenate']:
            col_off = 0
            for reader, reader_rows in zip(readers, rows):
                reader.rows_embedding_device(reader_rows, out=out, col_off=col_off)
                col_off += reader.dim
        else:
            for position, (reader, reader_rows) in enumerate(zip(readers, rows)):
                last = position == len(readers) - 1
                reader.rows_embedding_device(
                    reader_rows, out=out, accumulate=position > 0, divisor=float(len(readers)) if last else 0.0)
        return out

    def tokenizer_embedding(self, tokenizer):
        '''Merged results of tokenizer_embedding call from all readers
        Parameters
        ----------
        tokenizer : keras.preprocessing.text.Tokenizer
        '''
        return self.batch_embedding(tokenizer_word_list(tokenizer))

    def tokenizer_embedding_device(self, tokenizer):
        '''tokenizer_embedding merged on the GPU (see batch_embedding_device)'''
        return self.batch_embedding_device(tokenizer_word_list(tokenizer))
