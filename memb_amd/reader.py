"""Read side: the reference's `memb.Reader` interface (python/memb/reader.py) over
the HIP batch-lookup path. `reader[word]` / `reader[list_of_words]` return numpy
float32 exactly as the reference does; words the model does not know give zeros.
Everything below "additions" is new: row ids, strided outputs, results that stay
on the GPU."""
from abc import ABC, abstractmethod

import numpy as np

from . import _memb


class BaseReader(ABC):
    """What Reader, ReadersUnion and ShardedReader have in common: indexing and the
    exports built on batch_embedding."""

    def __getitem__(self, key):
        # a str selects one vector (1-D), a list a matrix (2-D); nothing else is accepted
        if isinstance(key, str):
            return self.word_embedding(key)
        if isinstance(key, list):
            return self.batch_embedding(key)
        raise TypeError('Key type is not supported')

    def to_keyed_vectors(self):
        """The whole model as a gensim KeyedVectors (one full-vocabulary lookup)"""
        try:
            from gensim.models import KeyedVectors
        except ImportError:
            raise ImportError('You must install gensim for KeyedVectors export')
        vocabulary = self.keys()
        exported = KeyedVectors(self.dim)
        exported.add(vocabulary, self.batch_embedding(vocabulary))
        return exported

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


def _current_stream(torch, index):
    '''torch's current stream on device `index` as a raw hipStream_t (the C call behind torch.cuda.current_stream,
    without building a Stream object: a microsecond per lookup)'''
    raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if raw is not None:
        try:
            return raw(index)
        except TypeError:   # (a private entry point: should its signature change, the public one still works)
            pass
    return torch.cuda.current_stream(index).cuda_stream


def tokenizer_word_list(tokenizer):
    """The words of a keras Tokenizer placed at their indices, '' where an index
    has no word (index 0 never has one). With `num_words` set only indices below
    it are kept, as the reference does (python/memb/reader.py:100-109); the
    embedding matrix of the tokenizer is then batch_embedding of this list."""
    entries = list(tokenizer.word_index.items())
    limit = tokenizer.num_words
    if limit is None:
        limit = max(index for _, index in entries) + 1
    else:
        entries = [(word, index) for word, index in entries if index < limit]
    slots = [''] * limit
    for word, index in entries:
        slots[index] = word
    return slots


class Reader(BaseReader):
    """One memb file, staged to a GPU on first use; lookups decode there.

    filename     str or path-like
    num_threads  host threads for the word search of large batches, 0 = one per core
    device       HIP device index (not in the reference API); default: environment
                 variable MEMB_HIP_DEVICE, else 0. 'cpu': decode on the host -- the reference's
                 own serial / threaded CPU path restated, for hosts without a GPU; it is only
                 ever used when asked for (a reader on a HIP device never falls back to it)
    host_below   host batches (words in, numpy out) of at most this many words are decoded on
                 the host although the reader lives on a GPU: a single word then costs no kernel
                 launch and no PCIe round trip. 0 = never (default, or MEMB_HOST_BELOW)
    max_direct_decode_bits
                 width of the first-level decode table, 0 = library default (results
                 never depend on it; the reference's tests force 1, src/tests.cpp:76-88)

    Footprint: the model itself (info()['device_bytes']) is staged on first use. Batches of 4096 words
    and more -- host results too -- are also SEARCHED on the device: the first such call copies the keys
    to HBM and builds a hash table over them there (16-byte slots, at least 2 x len(reader) of them, plus
    the keys: about 160 MB and some tens of milliseconds for a 2.2 M-word model; info()['word_index_bytes']).
    stage_words() pays that up front; device='cpu' and host_below keep a reader's host batches off it.
    """

    def __init__(self, filename, num_threads=0, device=None, max_direct_decode_bits=0, host_below=None):
        super().__init__()
        name = str(filename)
        if device in ('cpu', 'host'):
            device = _memb.HOST_DEVICE
        if device is None and not max_direct_decode_bits:
            self._impl = _memb.Reader(name, num_threads)
        else:
            self._impl = _memb.Reader(name, num_threads, -1 if device is None else int(device), max_direct_decode_bits)
        if host_below is not None:
            self._impl.set_host_below(int(host_below))
        self._word_batch = None   # packed query words of resolve_rows_device (pinned + device buffers, kept between calls)

    @property
    def dim(self):
        """length of every vector"""
        return self._impl.dim()

    @property
    def device(self):
        """HIP device index, or 'cpu' for a reader that decodes on the host"""
        index = self._impl.device()
        return 'cpu' if index == _memb.HOST_DEVICE else index

    @property
    def host_rows_decoded(self):
        """rows decoded by the host path so far (0 for a GPU reader with default settings)"""
        return self._impl.host_rows_decoded()

    def __len__(self):
        return self._impl.size()

    def keys(self):
        """all words of the model, sorted"""
        return self._impl.keys()

    def word_embedding(self, word):
        """float32 vector of shape (dim,); zeros for an unknown word"""
        return self._impl.word_embedding(word)

    def batch_embedding(self, words):
        """float32 matrix of shape (len(words), dim), one row per word in the given
        order; rows of unknown words are zeros"""
        return self._impl.batch_embedding(words)

    def tokenizer_embedding(self, tokenizer):
        """weights for an Embedding layer indexed like the keras Tokenizer"""
        return self.batch_embedding(tokenizer_word_list(tokenizer))

    def to_keyed_vectors(self):
        """The whole model as a gensim KeyedVectors. The reference looks every key up again (python/memb/reader.py:27-28:
        batch_embedding(keys())); keys() IS the row order, so the rows are decoded by number and nothing is searched
        (SURVEY 8f-1's fast path) -- the same matrix."""
        try:
            from gensim.models import KeyedVectors
        except ImportError:
            raise ImportError('You must install gensim for KeyedVectors export')
        vocabulary = self.keys()
        exported = KeyedVectors(self.dim)
        exported.add(vocabulary, self.rows_embedding(np.arange(len(vocabulary), dtype=np.uint32)))
        return exported

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

    def rows_embedding_device(self, rows, out=None, col_off=0, accumulate=False, divisor=0.0, order=None):
        '''Lookup that never leaves the GPU.
        Parameters
        ----------
        rows : torch.Tensor (int32 view of the uint32 row ids, on this reader's device)
        out : torch.Tensor float32 (n, >= col_off + dim) on the same device, optional
        accumulate : add the rows to what `out` holds instead of overwriting it
        divisor : if non-zero, divide the (accumulated) rows by it
        order : None, or 'random' -- a hint that the rows come in no particular order (token ids, shuffled keys): batches
            of more than 524 000 rows then keep blocks of four wavefronts, 3 % faster for such rows (key-order dumps like
            the default of eight). Never changes a result.
        '''
        # (a small batch is seven microseconds of which the kernel is three: every attribute is fetched once)
        import torch
        device = rows.device
        if device.type != 'cuda' or rows.dtype not in (torch.int32, torch.uint32) or not rows.is_contiguous():
            raise TypeError('rows must be a contiguous int32/uint32 tensor on the GPU')
        n = rows.numel()
        if out is None:
            out = torch.empty((n, col_off + self.dim), dtype=torch.float32, device=device)
        if out.dtype != torch.float32 or out.dim() != 2 or out.stride(1) != 1 or out.shape[0] != n:
            raise TypeError('out must be a float32 (n, width) tensor with unit column stride')
        # the kernel runs on this reader's device with these pointers: both tensors must live there
        index = device.index
        if index != self._impl.device() or out.device != device:
            raise ValueError('rows and out must be on cuda:{} (the device this reader is staged on), got {} and {}'.format(
                self.device, device, out.device))
        self._impl.rows_to_device(
            rows.data_ptr(), n, out.data_ptr(), out.stride(0), col_off, _current_stream(torch, index), accumulate, float(divisor),
            order == 'random')
        return out

    def stage_words(self):
        '''Copy the model's keys to the GPU and build the hash table over them there (once; resolve_rows_device does it
        on first use). After this, info()['device_bytes'] includes the word index.'''
        self._impl.stage_words()

    def resolve_rows_device(self, words, out=None):
        '''resolve_rows on the GPU: the words are packed into pinned memory by pooled host threads, copied once, and
        looked up by one kernel in a hash table over the model's keys (the same answers as the host search -- the
        reference's lower_bound + strcmp, src/trained_compression.cpp:115-125 -- misses as 0xFFFFFFFF). Returns a
        torch.int32 tensor on this reader's device; nothing waits for the GPU, the row ids never visit the host.
        out : optional contiguous int32 / uint32 tensor of len(words) entries on this reader's device'''
        import torch
        index = self._impl.device()
        if index == _memb.HOST_DEVICE:
            raise RuntimeError("this reader decodes on the host (device 'cpu'): the device word search needs a reader on a HIP device")
        n = len(words)
        if out is None:
            out = torch.empty((n,), dtype=torch.int32, device='cuda:{}'.format(index))
        elif (out.device.type != 'cuda' or out.device.index != index or out.dtype not in (torch.int32, torch.uint32)
              or not out.is_contiguous() or out.numel() != n):
            raise TypeError('out must be a contiguous int32/uint32 tensor of len(words) entries on cuda:{}'.format(index))
        if self._word_batch is None:
            self._word_batch = _memb.WordBatch(index)
        self._impl.words_to_rows_device(self._word_batch, words, out.data_ptr(), _current_stream(torch, index))
        return out

    def resolve_packed_device(self, data, offsets, out=None):
        '''resolve_rows_device for words that are packed already (a tokenizer's output): word i is the UTF-8 bytes
        data[offsets[i]:offsets[i + 1]]. No str object is touched -- the walk over a list of 2.2 M str is a cache miss per
        word and the larger half of resolve_rows_device's time.
        data : bytes-like (bytes, bytearray, memoryview, numpy.uint8) with offsets a numpy.uint32 array of n + 1 ascending
            entries -- copied once into pinned memory by pooled threads (GIL released), the lookups of finished runs overlap
            the copy of later ones; or BOTH torch tensors on this reader's device (uint8, int32 / uint32): looked up in place
            (memb_hip_resolve_packed_device).
        Returns the torch.int32 row ids on this reader's device (0xFFFFFFFF = not in the model).'''
        import torch
        index = self._impl.device()
        if index == _memb.HOST_DEVICE:
            raise RuntimeError("this reader decodes on the host (device 'cpu'): the device word search needs a reader on a HIP device")
        on_device = isinstance(data, torch.Tensor) or isinstance(offsets, torch.Tensor)
        n = int(offsets.numel() if isinstance(offsets, torch.Tensor) else len(offsets)) - 1
        if n < 0:
            raise ValueError('offsets needs n + 1 entries')
        if out is None:
            out = torch.empty((n,), dtype=torch.int32, device='cuda:{}'.format(index))
        elif (out.device.type != 'cuda' or out.device.index != index or out.dtype not in (torch.int32, torch.uint32)
              or not out.is_contiguous() or out.numel() != n):
            raise TypeError('out must be a contiguous int32/uint32 tensor of n entries on cuda:{}'.format(index))
        if on_device:
            if not (isinstance(data, torch.Tensor) and isinstance(offsets, torch.Tensor)):
                raise TypeError('data and offsets must both be torch tensors on the device, or both host buffers')
            for tensor, kinds in ((data, (torch.uint8,)), (offsets, (torch.int32, torch.uint32))):
                if tensor.device.type != 'cuda' or tensor.device.index != index or tensor.dtype not in kinds or not tensor.is_contiguous():
                    raise TypeError('device-resident words: contiguous uint8 bytes and int32 offsets on cuda:{}'.format(index))
            self._impl.packed_device_to_rows_device(data.data_ptr(), offsets.data_ptr(), n, out.data_ptr(), _current_stream(torch, index))
            return out
        if self._word_batch is None:
            self._word_batch = _memb.WordBatch(index)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        self._impl.packed_to_rows_device(self._word_batch, data, offsets, out.data_ptr(), _current_stream(torch, index))
        return out

    def batch_embedding_device(self, words):
        '''batch_embedding with the result left on the GPU as a torch.Tensor (DLPack capable). Words are resolved on
        the GPU as well (resolve_rows_device): the only host work is packing the strings.'''
        return self.rows_embedding_device(self.resolve_rows_device(words))

    def rows_embedding_device_many(self, batches):
        '''Several lookups in ONE kernel launch (memb_hip_decode_batches_device): `batches` is a sequence of
        (rows, out) or (rows, out, col_off) with the tensors rows_embedding_device takes; results are those of one
        rows_embedding_device call per entry. For serving loops whose batches are too small to fill the GPU: launch gap,
        prologue and tail are paid once. Returns the list of `out` tensors.'''
        import torch
        index = self._impl.device()
        if index == _memb.HOST_DEVICE:
            raise RuntimeError("this reader decodes on the host (device 'cpu'): device buffers need a reader on a HIP device")
        descriptors = []
        outs = []
        for entry in batches:
            rows, out = entry[0], entry[1]
            col_off = entry[2] if len(entry) > 2 else 0
            if rows.device.type != 'cuda' or rows.dtype not in (torch.int32, torch.uint32) or not rows.is_contiguous():
                raise TypeError('rows must be a contiguous int32/uint32 tensor on the GPU')
            n = rows.numel()
            if out.dtype != torch.float32 or out.dim() != 2 or out.stride(1) != 1 or out.shape[0] != n:
                raise TypeError('out must be a float32 (n, width) tensor with unit column stride')
            if rows.device.index != index or out.device != rows.device:
                raise ValueError('rows and out must be on cuda:{} (the device this reader is staged on)'.format(self.device))
            if out.shape[1] < col_off + self.dim:
                raise ValueError('out is narrower than col_off + dim')
            descriptors.append((rows.data_ptr(), n, out.data_ptr(), out.stride(0) if n > 1 else out.shape[1], col_off))
            outs.append(out)
        self._impl.batches_to_device(descriptors, _current_stream(torch, index))
        return outs

    def tokenizer_embedding_device(self, tokenizer):
        '''tokenizer_embedding with the weights left on the GPU: a torch.Tensor that
        torch.nn.Embedding.from_pretrained (or any DLPack consumer) takes as is, so the
        embedding matrix of a model never crosses PCIe'''
        return self.batch_embedding_device(tokenizer_word_list(tokenizer))

    def info(self, batch_words=0):
        '''Facts about the device context (stages the model on first call). The kernel and its launch geometry
        are chosen by batch size: batch_words names the size they are reported for (0 = a large batch).'''
        return self._impl.info(int(batch_words))

    def set_option(self, name, value):
        '''Tuning knob of the device context ('persistent', 'tiles_per_wave', 'waves_per_block', 'union_split', 'union_fused',
        'host_expand': include/memb_hip.h, memb_hip_ctx_set_option); results never depend on them'''
        self._impl.set_option(str(name), int(value))
