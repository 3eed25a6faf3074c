from abc import ABC, abstractmethod

import numpy as np

from . import _memb


class BaseReader(ABC):
    def __getitem__(self, key):
        '''Obtain vector representation for a word or a list of words
        (reference python/memb/reader.py:6-17)
        Parameters
        ----------
        key: str of list of str
        '''
        if isinstance(key, str):
            return self.word_embedding(key)
        elif isinstance(key, list):
            return self.batch_embedding(key)
        else:
            raise TypeError('Key type is not supported')

    def to_keyed_vectors(self):
        '''Export model content to KeyedVectors object (reference python/memb/reader.py:19-30)'''
        try:
            from gensim.models import KeyedVectors
        except ImportError:
            raise ImportError('You must install gensim for KeyedVectors export')

        keyed_vectors = KeyedVectors(self.dim)
        words = self.keys()
        keyed_vectors.add(words, self.batch_embedding(words))

        return keyed_vectors

    @abstractmethod
    def keys(self):
        pass

    @abstractmethod
    def word_embedding(self, word):
        pass

    @abstractmethod
    def batch_embedding(self, words):
        pass

    @abstractmethod
    def tokenizer_embedding(self, tokenzer):
        pass


def tokenizer_word_list(tokenizer):
    '''Words of a keras Tokenizer laid out by index, '' in unused slots
    (reference python/memb/reader.py:100-109)'''
    word_indices = tokenizer.word_index.items()
    if tokenizer.num_words is not None:
        word_indices = [item for item in word_indices if item[1] < tokenizer.num_words]
        max_index = tokenizer.num_words
    else:
        max_index = max([item[1] for item in word_indices]) + 1

    sorted_word_list = [''] * max_index
    for word, idx in word_indices:
        sorted_word_list[idx] = word
    return sorted_word_list


class Reader(BaseReader):
    '''Reader object allows to obtain embeddings for requested words quickly,
    decoding them on the GPU on the fly (reference python/memb/reader.py:49-111)
    Parameters
    ----------
    filename : str or pathib.Path
    num_threads : int
        Number of host threads used to look up large batches of words.
        Pass 0 to use as much threads as there are cores in the system
    device : int, optional
        HIP device that holds the model and runs the lookups (not in the
        reference API). Default: environment variable MEMB_HIP_DEVICE, else 0
    Attributes
    ----------
    dim : int
        Embeddings dimension
    '''

    def __init__(self, filename, num_threads=0, device=None, max_direct_decode_bits=0):
        super().__init__()
        if device is None and not max_direct_decode_bits:
            self._impl = _memb.Reader(str(filename), num_threads)
        else:
            self._impl = _memb.Reader(
                str(filename), num_threads, -1 if device is None else int(device), max_direct_decode_bits)

    @property
    def dim(self):
        return self._impl.dim()

    @property
    def device(self):
        return self._impl.device()

    def __len__(self):
        return self._impl.size()

    def keys(self):
        '''List of words contained in model'''
        return self._impl.keys()

    def word_embedding(self, word):
        '''Obtain one-dimensional array of type float32 for a given word.
        If word is not present in the model, array filled with zeros is returned
        Parameters
        ----------
        word : str
        '''
        return self._impl.word_embedding(word)

    def batch_embedding(self, words):
        '''Obtain two-dimensional array of type float32 for a given list of words.
        Positions for words not present in the model are filled with zeros
        Parameters
        ----------
        words : list of str
        '''
        return self._impl.batch_embedding(words)

    def tokenizer_embedding(self, tokenizer):
        '''Convert keras.preprocessing.text.Tokenizer to weights of Embedding layer
        Parameters
        ----------
        tokenizer : keras.preprocessing.text.Tokenizer
        '''
        return self.batch_embedding(tokenizer_word_list(tokenizer))

    # ---- additions: row ids and device-resident results ----

    def resolve_rows(self, words):
        '''Row ids (positions in sorted key order) as numpy.uint32;
        0xFFFFFFFF marks words that are not in the model'''
        return self._impl.resolve_rows(words)

    def rows_embedding(self, rows):
        '''batch_embedding for already resolved row ids'''
        return self._impl.rows_embedding(np.ascontiguousarray(rows, dtype=np.uint32))

    def rows_embedding_into(self, rows, out, col_off=0):
        '''rows_embedding into columns [col_off, col_off + dim) of a C-contiguous float32 matrix
        (or a row range of one: slices of a shared result can be filled from several threads)'''
        self._impl.rows_embedding_into(np.ascontiguousarray(rows, dtype=np.uint32), out, col_off)

    def batch_embedding_into(self, words, out, col_off=0):
        '''Write the batch into columns [col_off, col_off + dim) of a wider float32 matrix'''
        self._impl.batch_embedding_into(words, out, col_off)

    def rows_embedding_device(self, rows, out=None, col_off=0, accumulate=False, divisor=0.0):
        '''Lookup that never leaves the GPU.
        Parameters
        ----------
        rows : torch.Tensor (int32 view of the uint32 row ids, on this reader's device)
        out : torch.Tensor float32 (n, >= col_off + dim) on the same device, optional
        accumulate : add the rows to what `out` holds instead of overwriting it
        divisor : if non-zero, divide the (accumulated) rows by it
        '''
        import torch
        if rows.device.type != 'cuda' or rows.dtype not in (torch.int32, torch.uint32) or not rows.is_contiguous():
            raise TypeError('rows must be a contiguous int32/uint32 tensor on the GPU')
        n = rows.numel()
        if out is None:
            out = torch.empty((n, col_off + self.dim), dtype=torch.float32, device=rows.device)
        if out.dtype != torch.float32 or out.dim() != 2 or out.stride(1) != 1 or out.shape[0] != n:
            raise TypeError('out must be a float32 (n, width) tensor with unit column stride')
        stream = torch.cuda.current_stream(rows.device).cuda_stream
        self._impl.rows_to_device(
            rows.data_ptr(), n, out.data_ptr(), out.stride(0), col_off, stream, accumulate, float(divisor))
        return out

    def batch_embedding_device(self, words):
        '''batch_embedding with the result left on the GPU as a torch.Tensor (DLPack capable)'''
        import torch
        rows = torch.from_numpy(self.resolve_rows(words).view('int32')).to('cuda:{}'.format(self.device))
        return self.rows_embedding_device(rows)

    def tokenizer_embedding_device(self, tokenizer):
        '''tokenizer_embedding with the weights left on the GPU: a torch.Tensor that
        torch.nn.Embedding.from_pretrained (or any DLPack consumer) takes as is, so the
        embedding matrix of a model never crosses PCIe'''
        return self.batch_embedding_device(tokenizer_word_list(tokenizer))

    def info(self):
        '''Facts about the device context (stages the model on first call)'''
        return self._impl.info()
