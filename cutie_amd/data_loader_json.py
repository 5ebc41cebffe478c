"""Host-side mirror of the reference's grid-table producer, with the token -> grid mapping on
the GPU.

Same constructor, attributes, method names and batch dict as ``DataLoader`` in
/root/reference/data_loader_json.py:34-312, so the reference's drivers
(main_train_json.py:134-147,228-257; main_evaluate_json.py:49-58) can use this class
unchanged.  What differs underneath:

  * ``_positional_mapping`` (:437-577, the per-step hot function) packs the documents' tokens
    into four int32 arrays and calls the HIP kernel ``cutie_grid_build`` through the C ABI.
    There is no CPU fallback: without libcutie_hip.so / a GPU the call raises CutieError.
    Besides the reference's numpy outputs the batch dict carries the device tensors
    (``batch['device']``) so a training step never re-uploads what the kernel just wrote.
  * ``dict/*.npy`` are read with cutie_amd.dictfile (no unpickling) and written in the same
    format (:79-81, :107-110).
  * load-time preparation (``_collect_data`` :704-831, ``_dress_text`` :887-925,
    ``_dress_bbox`` :927-935, ``_cal_rows_cols`` :660-694) is restated in plain Python with the
    same integer / double arithmetic; it is pinned by tests/golden/faktura1_items.json and
    shapes.json, which the reference's own loader produced.

positional_mapping_strategy 2 (the default of the reference's drivers) runs too: its "expansion"
never triggers - the search for a free slot starts on the cell the strategy-1 shift has just
freed - so it is strategy 1 (tests/golden/strategy2_cases.json holds the reference's own verdict).

Not carried over (out of scope): cv2 image loading for CUTIE2 (``_positional_sampling`` takes
already-decoded images through ``image_reader``), the grid visualiser and the unused
``_form_label_matrix`` / ``build_gt_pyramid`` stubs.
"""
import json
import os
import random
import unicodedata
from collections import defaultdict

import numpy as np

from . import dictfile, tokenization

DEFAULT_CLASSES = ['O', 'DATE_SALE', 'ADDRESS_SELLER', 'DOC_NR', 'DATE_CREATION', 'ADDRESS_CONTRACTOR',
                   'VAT_ID_CONTRACTOR', 'PAYMENT_METHOD', 'VAT_ID_SELLER', 'PAYMENT_BANK_NR', 'TOTAL_PAY',
                   'TOTAL_CURRENCY', 'TOTAL_TAX', 'TOTAL_WITH_TAX', 'TOTAL_WITHOUT_TAX', 'DATE_PAYMENT',
                   'NAME_SELLER', 'NAME_CONTRACTOR']


def is_number(s):
    """data_loader_json.py:20-32: float() parses it, or it is one numeric unicode character"""
    try:
        float(s)
        return True
    except ValueError:
        pass
    try:
        unicodedata.numeric(s)
        return True
    except (TypeError, ValueError):
        return False


def _opt(params, name, default):
    return getattr(params, name) if hasattr(params, name) else default


class _Prefetcher(object):
    def __init__(self, fn, depth, device=None):
        import queue
        import threading
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._fn = fn
        self._device = device      # the creating thread's GPU: a new thread starts with device 0 current
        self._t = threading.Thread(target=self._run, name='cutie-prefetch', daemon=True)
        self._t.start()

    def _run(self):
        import queue
        if self._device is not None:
            import torch
            torch.cuda.set_device(self._device)
        while not self._stop.is_set():
            try:
                item = ('ok', self._fn())
            except BaseException as e:          # hand the error to the consumer, then stop
                item = ('err', e)
            while not self._stop.is_set():
                try:
                    self._q.put(item, timeout=0.1)
                    break
                except queue.Full:
                    continue
            if item[0] == 'err' or item[1] is None:     # fetch_test_data ends a pass with None
                return

    def __iter__(self):
        return self

    def __next__(self):
        if self._stop.is_set():
            raise StopIteration
        kind, val = self._q.get()
        if kind == 'err':
            self.close()
            raise val
        if val is None:
            self.close()
            raise StopIteration
        return val

    def close(self):
        self._stop.set()
        self._t.join(timeout=5.0)


_FRONT = None      # per worker process: (loader clone, update_dict)


def _front_init(state, known_words, update_dict):
    """worker set-up: a DataLoader clone that holds only what _collect_label / _collect_data read; its dictionary
    starts with the parent's words at count 0 so that what a file adds can be told apart"""
    global _FRONT
    clone = object.__new__(DataLoader)
    clone.__dict__.update(state)
    clone.dictionary = dict.fromkeys(known_words, 0)
    _FRONT = (clone, update_dict)


def _front_file(path):
    """one file in a worker -> (label, item lists, word count deltas, words new to the parent in first-use order)"""
    clone, update_dict = _FRONT
    n_before = len(clone.dictionary)
    label, docs = clone._load_file(path, update_dict)
    counts = {}
    for w, c in clone.dictionary.items():
        if c:
            counts[w] = c
            clone.dictionary[w] = 0
    new_words = list(clone.dictionary)[n_before:]      # _dress_text appended them in first-use order (dicts keep it)
    return label, docs, counts, new_words


class DataLoader(object):
    """grid tables producer (reference: data_loader_json.py:34)"""

    def __init__(self, params, update_dict=True, load_dictionary=False, data_split=0.75, device=None,
                 verbose=True):
        self.random = False
        self.data_laundry = False
        self.encoding_factor = 1
        self.classes = list(_opt(params, 'classes', DEFAULT_CLASSES))
        self.device = device          # torch device of the grid builder; None = current GPU
        import threading
        self._tls = threading.local()     # .last_device: per thread, so a prefetch thread and a foreground fetch_* do not clobber each other
        self.verbose = verbose

        self.doc_path = params.doc_path
        self.doc_test_path = params.test_path
        self.use_cutie2 = params.use_cutie2
        self.text_case = params.text_case
        self.tokenize = params.tokenize
        if self.tokenize:
            # the reference hard-codes the cwd-relative 'dict/vocab.txt' (:54)
            self.tokenizer = tokenization.FullTokenizer(_opt(params, 'vocab_file', 'dict/vocab.txt'),
                                                        do_lower_case=not self.text_case)

        self.rows = self.encoding_factor
        self.cols = self.encoding_factor
        self.segment_grid = _opt(params, 'segment_grid', False)
        self.augment_strategy = _opt(params, 'augment_strategy', 1)
        self.pm_strategy = _opt(params, 'positional_mapping_strategy', 2)
        self.rows_segment = _opt(params, 'rows_segment', 72)
        self.cols_segment = _opt(params, 'cols_segment', 72)
        self.rows_target = _opt(params, 'rows_target', 64)
        self.cols_target = _opt(params, 'cols_target', 64)
        self.rows_ulimit = _opt(params, 'rows_ulimit', 80)
        self.cols_ulimit = _opt(params, 'cols_ulimit', 80)
        self.fill_bbox = _opt(params, 'fill_bbox', False)
        self.data_augmentation_dropout = _opt(params, 'data_augmentation_dropout', False)
        self.data_augmentation_extra = _opt(params, 'data_augmentation_extra', False)
        self.da_extra_rows = _opt(params, 'data_augmentation_extra_rows', 0)
        self.da_extra_cols = _opt(params, 'data_augmentation_extra_cols', 0)

        self.load_dictionary = load_dictionary
        self.dict_path = params.load_dict_from_path if load_dictionary else params.dict_path
        if self.load_dictionary:
            self.dictionary, self.word_to_index, self.index_to_word = \
                dictfile.load_dictionary_triplet(self.dict_path)
        else:
            self.dictionary = {'[PAD]': 0, '[UNK]': 0}
            self.word_to_index = {}
            self.index_to_word = {}

        self.data_split = data_split
        self.data_mode = 2            # 0: key and value are two classes, 1: one class, 2: value only
        self.remove_lowfreq_words = False
        self.num_classes = len(self.classes)
        self.batch_size = _opt(params, 'batch_size', 1)
        self.special_dict = {'*', '='}
        self.image_reader = _opt(params, 'image_reader', None)   # callable(path) -> HxWx3 array or None
        self.load_workers = _opt(params, 'load_workers', None)   # processes for the load-time front half; None = auto

        self.training_doc_files = self._get_filenames(self.doc_path)
        self.training_docs, self.training_labels = self.load_data(self.training_doc_files, update_dict=update_dict)

        self.num_words = len(self.dictionary)
        self._updae_word_to_index()
        self._update_docs_dictionary(self.training_docs, 3, self.remove_lowfreq_words)

        if _opt(params, 'save_dict', True):      # the reference always writes them back (:107-110)
            dictfile.save_npy_dict(self.dict_path + '_dictionary.npy', self.dictionary)
            dictfile.save_npy_dict(self.dict_path + '_word_to_index.npy', self.word_to_index)
            dictfile.save_npy_dict(self.dict_path + '_index_to_word.npy', self.index_to_word)
            np.save(self.dict_path + '_classes.npy', self.classes)

        num_training = int(len(self.training_docs) * self.data_split)
        everything = list(range(len(self.training_docs)))
        chosen = everything[:num_training]
        if self.random:
            chosen = random.sample(everything, num_training)
        rest = list(set(everything).difference(set(chosen)))
        self.validation_docs = [self.training_docs[x] for x in rest]
        self.training_docs = [self.training_docs[x] for x in chosen]
        self.validation_labels = self.training_labels
        self._say('\n\nDATASET: %d vocabularies, %d target classes' % (len(self.dictionary), len(self.classes)))
        self._say('DATASET: %d for training, %d for validation' % (len(self.training_docs), len(self.validation_docs)))

        self.test_doc_files = self._get_filenames(params.test_path) if hasattr(params, 'test_path') else []
        self.test_docs, self.test_labels = self.load_data(self.test_doc_files, update_dict=update_dict)
        self._say('DATASET: %d for test from %s \n' % (len(self.test_docs), _opt(params, 'test_path', '_')))

        self.data_shape_statistic()
        if len(self.training_docs) > 0:
            self.rows, self.cols, _, _ = self._cal_rows_cols(self.training_docs)
            self._say('\nDATASHAPE: data set with maximum grid table of ({},{}), updated.\n'.format(self.rows, self.cols))
        else:
            self.rows, self.cols = self.rows_ulimit, self.cols_ulimit

        self.training_data_tobe_fetched = list(range(len(self.training_docs)))
        self.validation_data_tobe_fetched = list(range(len(self.validation_docs)))
        self.test_data_tobe_fetched = list(range(len(self.test_docs)))

    def _say(self, msg):
        if self.verbose:
            print(msg)

    # ------------------------------------------------------------------ dictionary upkeep
    def _updae_word_to_index(self):      # (sic) the reference's spelling, :144
        if self.load_dictionary:
            max_index = len(self.word_to_index.keys())
            for word in self.dictionary:
                if word not in self.word_to_index:
                    max_index += 1
                    self.word_to_index[word] = max_index
                    self.index_to_word[max_index] = word
        else:
            words = list(self.dictionary.keys())
            self.word_to_index = {w: i for i, w in enumerate(words)}
            self.index_to_word = {i: w for i, w in enumerate(words)}

    def _update_docs_dictionary(self, docs, lower_limit, remove_lowfreq_words):
        """:156-167. The reference rebinds a loop variable, so the documents themselves are
        never changed; only the word counts move.  Kept, counts included."""
        if remove_lowfreq_words:
            for doc in docs:
                for line in doc:
                    word = line[1]
                    if self.dictionary[word] < lower_limit:
                        self.dictionary[word] -= 1
                        self.dictionary['[UNK]'] += 1

    # ------------------------------------------------------------------ batches (a3)
    def _map_with_expand(self, docs, labels, rows, cols, what):
        out = self._positional_mapping(docs, labels, rows, cols)
        if out[6] > cols:          # cannot happen: no strategy grows the grid (see the module docstring)
            raise NotImplementedError('%s grid grew from %d to %d columns' % (what, cols, out[6]))
        return out

    def _batch(self, mapped, images, ps_1d_indices, rows, cols):
        grid_table, gt_classes, bboxes, label_mapids, bbox_mapids, file_names, _, _, _ = mapped
        batch = {'grid_table': np.array(grid_table), 'gt_classes': np.array(gt_classes),
                 'data_image': np.array(images), 'ps_1d_indices': np.array(ps_1d_indices),
                 'bboxes': bboxes, 'label_mapids': label_mapids, 'bbox_mapids': bbox_mapids,
                 'file_name': file_names, 'shape': [rows, cols]}
        batch['device'] = self._tls.last_device       # {'grid_table': int32 [B,H,W,1], 'gt_classes': int8 [B,H,W]}
        return batch

    def next_batch(self):
        """:169-208"""
        batch_size = self.batch_size
        while True:
            if len(self.training_data_tobe_fetched) < batch_size:
                self.training_data_tobe_fetched = list(range(len(self.training_docs)))
            selected_index = random.sample(self.training_data_tobe_fetched, batch_size)
            self.training_data_tobe_fetched = list(set(self.training_data_tobe_fetched).difference(set(selected_index)))
            training_docs = [self.training_docs[x] for x in selected_index]

            rows, cols, pre_rows, pre_cols = self._cal_rows_cols(
                training_docs, extra_augmentation=self.data_augmentation_extra, dropout=self.data_augmentation_dropout)
            if self.data_augmentation_extra:
                self._say('Training grid AUGMENT size: ({},{}) from ({},{})'.format(rows, cols, pre_rows, pre_cols))
            mapped = self._map_with_expand(training_docs, self.training_labels, rows, cols, 'Training')

            images, ps_1d_indices = [], []
            if self.use_cutie2:
                images, ps_1d_indices = self._positional_sampling(self.doc_path, mapped[5], mapped[7], mapped[8], mapped[6])
                if len(images) == batch_size:
                    break
            else:
                break
        return self._batch(mapped, images, ps_1d_indices, rows, cols)

    def prefetch(self, depth=2, fetch='next_batch'):
        """Iterator over `fetch()` results produced `depth` batches ahead on a background thread, so
        that the host half of a batch (shape choice, packing, the read-back the evaluation dicts need)
        overlaps the device step instead of serialising with it as main_train_json.py:228 does.
        Batches come out in the order the thread drew them; call .close() (or exhaust it) to stop."""
        import torch
        dev = torch.device(self.device) if self.device is not None else (
            torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None)
        return _Prefetcher(getattr(self, fetch), depth, dev)

    def fetch_validation_data(self):
        """:210-272 (one document; the reference takes max(rows_target, .) for both axes, :223-224)"""
        while True:
            if len(self.validation_data_tobe_fetched) == 0:
                self.validation_data_tobe_fetched = list(range(len(self.validation_docs)))
            selected_index = random.sample(self.validation_data_tobe_fetched, 1)
            self.validation_data_tobe_fetched = list(set(self.validation_data_tobe_fetched).difference(set(selected_index)))
            validation_docs = [self.validation_docs[x] for x in selected_index]

            real_rows, real_cols, _, _ = self._cal_rows_cols(validation_docs, extra_augmentation=False)
            rows = max(self.rows_target, real_rows)
            cols = max(self.rows_target, real_cols)
            mapped = self._map_with_expand(validation_docs, self.validation_labels, rows, cols, 'Validation')

            images, ps_1d_indices = [], []
            if self.use_cutie2:
                images, ps_1d_indices = self._positional_sampling(self.doc_path, mapped[5], mapped[7], mapped[8], mapped[6])
                if len(images) == 1:
                    break
            else:
                break
        return self._batch(mapped, images, ps_1d_indices, rows, cols)

    def fetch_test_data(self):
        """:274-312; returns None once per pass over the test set"""
        while True:
            if len(self.test_data_tobe_fetched) == 0:
                self.test_data_tobe_fetched = list(range(len(self.test_docs)))
                return None
            selected_index = self.test_data_tobe_fetched[0]
            self.test_data_tobe_fetched = list(set(self.test_data_tobe_fetched).difference(set([selected_index])))
            test_docs = [self.test_docs[selected_index]]

            real_rows, real_cols, _, _ = self._cal_rows_cols(test_docs, extra_augmentation=False)
            rows = max(self.rows_target, real_rows)
            cols = max(self.cols_target, real_cols)
            mapped = self._map_with_expand(test_docs, self.test_labels, rows, cols, 'Test')

            images, ps_1d_indices = [], []
            if self.use_cutie2:
                images, ps_1d_indices = self._positional_sampling(self.doc_test_path, mapped[5], mapped[7], mapped[8], mapped[6])
                if len(images) == 1:
                    break
            else:
                break
        return self._batch(mapped, images, ps_1d_indices, rows, cols)

    def data_shape_statistic(self):
        """:374-435"""
        def shape_statistic(docs):
            res_all, res_row, res_col = defaultdict(int), defaultdict(int), defaultdict(int)
            for doc in docs:
                rows, cols, _, _ = self._cal_rows_cols([doc])
                res_all[rows] += 1
                res_all[cols] += 1
                res_row[rows] += 1
                res_col[cols] += 1
            srt = lambda d: sorted(d.items(), key=lambda x: x[0], reverse=True)
            return srt(res_all), srt(res_row), srt(res_col)

        def report(suffix=''):
            for title, docs in (('Training', self.training_docs), ('\nValidation', self.validation_docs),
                                ('\nTest', self.test_docs)):
                a, r, c = shape_statistic(docs)
                self._say("%s statistic%s:  %s" % (title, suffix, a))
                self._say("\t num:  %d" % len(docs))
                self._say("\t rows statistic:  %s" % (r,))
                self._say("\t cols statistic:  %s" % (c,))

        report()
        if self.data_laundry:
            def launder(docs):
                docs[:] = [d for d in docs
                           if not (self._cal_rows_cols([d])[0] > self.rows_ulimit or
                                   self._cal_rows_cols([d])[1] > self.cols_ulimit)]
            self._say("\nRemoving grids with shape larger than ({},{}).".format(self.rows_ulimit, self.cols_ulimit))
            launder(self.training_docs)
            launder(self.validation_docs)
            report(' after laundary')

    # ------------------------------------------------------------------ a2: the hot function
    def pack_documents(self, docs, labels):
        """Flatten documents into the int32 arrays ``cutie_grid_build`` takes.

        Per token (data_loader_json.py:462-470,:510): box, dict id = word_to_index[text],
        entity_id*num_classes + class_id; per document its extent (left, top, right, bottom)."""
        packed = [self._packed_doc(doc, labels) for doc in docs]
        doc_off = np.zeros((len(docs) + 1,), dtype=np.int32)
        doc_off[1:] = np.cumsum([len(p[1]) for p in packed])
        cat = lambda k, shape, : (np.concatenate([p[k] for p in packed]) if packed else np.zeros(shape, dtype=np.int32))
        tok_box, tok_id, tok_ecid, tok_cls = cat(0, (0, 4)), cat(1, (0,)), cat(2, (0,)), cat(3, (0,))
        extent = np.stack([p[4] for p in packed]) if packed else np.zeros((0, 4), dtype=np.int32)
        return tok_box, tok_id, tok_ecid, tok_cls, extent, doc_off

    def _packed_doc(self, doc, labels):
        """int32 arrays of one document, built once: a document is mapped again every time it is
        drawn (and at every augmented shape), its tokens, dictionary ids and classes do not change
        after load time (the per-token Python loop was the host cost of a batch)."""
        cache = self.__dict__.setdefault('_pack_cache', {})
        key = (id(doc), id(labels))
        hit = cache.get(key)
        if hit is not None and hit[0] is doc and hit[1] is labels:
            return hit[2]
        n = len(doc)
        box = np.zeros((n, 4), dtype=np.int32)
        tid = np.zeros((n,), dtype=np.int32)
        ecid = np.zeros((n,), dtype=np.int32)
        cls = np.zeros((n,), dtype=np.int32)
        ext = np.zeros((4,), dtype=np.int32)
        for t, item in enumerate(doc):
            file_name, text, word_id = item[0], item[1], item[2]
            box[t] = item[3]
            tid[t] = self.word_to_index[text]
            entity_id, class_id = self._dress_class(file_name, word_id, labels)
            ecid[t] = entity_id * self.num_classes + class_id
            cls[t] = class_id
            ext[:] = item[4]             # the reference uses each item's own copy; they are equal
        packed = (box, tid, ecid, cls, ext)
        cache[key] = (doc, labels, packed)
        return packed

    def _positional_mapping(self, docs, labels, rows, cols):
        """:437-577 on the GPU.  Returns the reference's 9-tuple
        (grid_tables, grid_labels, bboxes, label_mapids, bbox_mapids, file_names, cols,
        ps_indices_x, ps_indices_y)."""
        import torch

        from . import ops

        if self.pm_strategy not in (0, 1, 2):
            raise ValueError('positional_mapping_strategy %r: 0 (skip), 1 (shift) or 2 (expand)' % (self.pm_strategy,))
        dev = torch.device(self.device) if self.device is not None else torch.device('cuda', torch.cuda.current_device())
        tok_box, tok_id, tok_ecid, tok_cls, extent, doc_off = self.pack_documents(docs, labels)
        up = lambda a: torch.from_numpy(a).to(dev, non_blocking=True)
        res = ops.grid_build(up(tok_box), up(tok_id), up(tok_ecid), up(extent), up(doc_off), rows, cols, self.pm_strategy)
        status = res['status'].cpu().numpy()
        if status.any():
            bad = int(np.nonzero(status)[0][0])
            raise IndexError('document %d (%s): a token maps outside the %dx%d grid'
                             % (bad, docs[bad][0][0] if docs[bad] else '?', rows, cols))
        grid = res['grid'].cpu().numpy()
        lab = res['labels'].cpu().numpy()
        ps_x = res['ps_x'].cpu().numpy()
        ps_y = res['ps_y'].cpu().numpy()
        order = res['order'].cpu().numpy()
        place = res['place'].cpu().numpy()
        self._tls.last_device = {'grid_table': res['grid'].unsqueeze(-1), 'gt_classes': res['labels'],
                             'ps_indices_x': res['ps_x'], 'ps_indices_y': res['ps_y']}

        grid_tables, grid_labels, ps_indices_x, ps_indices_y = [], [], [], []
        bboxes, label_mapids, bbox_mapids, file_names = {}, [], [], []
        for b, doc in enumerate(docs):
            lo, hi = int(doc_off[b]), int(doc_off[b + 1])
            # bbox_id is the 1-based reading-order index (:472); label_mapid is filled in reading
            # order (:489), bbox_mapid / bbox in placement order (:528-529, :556-557)
            label_mapid = [[] for _ in range(self.num_classes)]
            for k in range(lo, hi):
                label_mapid[int(tok_cls[k])].append(k - lo + 1)
            bbox = [[] for _ in range(2 * cols * rows)]
            bbox_mapid = {}
            for k in range(lo, hi):
                r, c = int(place[k, 0]), int(place[k, 1])
                if r < 0:
                    if self.verbose:
                        print('overlap in {} row {} r{}c{}!'.format(doc[0][0], '?', rows, cols))
                    continue
                idx = int(order[k])
                x_left, y_top, x_right, y_bottom = (int(v) for v in tok_box[lo + idx])
                bbox_mapid[r * cols + c] = idx + 1
                bbox[r * cols + c] = [x_left, y_top, x_right - x_left, y_bottom - y_top]
            file_name = doc[-1][0] if doc else None    # the loop variable's last value (:572)
            grid_tables.append(grid[b][..., None])
            grid_labels.append(lab[b])
            ps_indices_x.append(ps_x[b])
            ps_indices_y.append(ps_y[b])
            bboxes[file_name] = bbox
            label_mapids.append(label_mapid)
            bbox_mapids.append(bbox_mapid)
            file_names.append(file_name)
        return grid_tables, grid_labels, bboxes, label_mapids, bbox_mapids, file_names, cols, ps_indices_x, ps_indices_y

    def _positional_sampling(self, path, file_names, ps_indices_x, ps_indices_y, updated_cols):
        """:579-630 with the decoder injected: ``params.image_reader(path)`` returns an HxWx3
        array already at width ``updated_cols`` scale is NOT assumed - the resize is nearest
        neighbour on the host (cv2 is absent; SURVEY.md a14 replaces real images by synthetic
        ones).  ps_1d = floor(x*f) + floor(y*f)*W, clamped to W*h-2 (:599-603)."""
        if self.image_reader is None:
            raise NotImplementedError('CUTIE2 image input needs params.image_reader (cv2 is not available)')
        images, ps_1d_indices = [], []
        max_h, max_w = 0, updated_cols
        for i, file_name in enumerate(file_names):
            image = self.image_reader(os.path.join(path, file_name))
            if image is None:
                self._say('{} ignored due to image file not found.'.format(os.path.join(path, file_name)))
                break
            ps_1d_x = np.array(ps_indices_x[i], dtype=np.float32).reshape([-1])
            ps_1d_y = np.array(ps_indices_y[i], dtype=np.float32).reshape([-1])
            h, w = image.shape[0], image.shape[1]
            factor = max_w / w
            h = int(h * factor)
            ps_1d_x *= factor
            ps_1d_y *= factor
            ps_1d = np.int32(np.floor(ps_1d_x) + np.floor(ps_1d_y) * max_w)
            np.minimum(ps_1d, max_w * h - 2, out=ps_1d)
            yy = np.minimum((np.arange(h) / factor).astype(np.int64), image.shape[0] - 1)
            xx = np.minimum((np.arange(max_w) / factor).astype(np.int64), image.shape[1] - 1)
            image = (image[yy][:, xx].astype(np.float64) - 127.5) / 255
            ps_1d_indices.append(ps_1d)
            images.append(image)
            max_h = max(max_h, h)
        for i, image in enumerate(images):
            pad_img = np.zeros([max_h, max_w, 3], dtype=image.dtype)
            pad_img[:image.shape[0], :, :] = image
            images[i] = pad_img
        return images, ps_1d_indices

    # ------------------------------------------------------------------ a1: load-time preparation
    def _load_file(self, file, update_dict):
        """one document file -> (label dict or None, list of item lists): the per-file body of :632-658"""
        with open(file, encoding='utf-8') as f:
            data = json.load(f)
        file_id = data['global_attributes']['file_id']
        label = self._collect_label(file_id, data['fields'])
        if not label:            # corrupted sample
            return None, []
        return label, self._collect_data(file_id, data['text_boxes'], update_dict)

    def load_data(self, data_files, update_dict=False):
        """:632-658.  The reference walks the files one by one; here the per-file work (JSON parse, WordPiece,
        the natural-shape search of _collect_data) goes to `load_workers` processes when there are enough files,
        and the results are merged in FILE ORDER, dictionary counts and new words included, so the outcome is the
        serial one (tests/test_data_loader.py pins both against the reference's own items)."""
        label_dressed, doc_dressed = {}, []
        if not data_files:
            self._say("no data file found.")
        workers = self._pool_size(len(data_files))
        if workers <= 1:
            for file in data_files:
                label, docs = self._load_file(file, update_dict)
                if label:
                    label_dressed.update(label)
                    doc_dressed.extend(docs)
            return doc_dressed, label_dressed
        import multiprocessing as mp
        state = {k: v for k, v in self.__dict__.items() if k not in ('_tls', '_pack_cache', 'image_reader', 'dictionary')}
        known = list(self.dictionary)
        chunk = max(1, len(data_files) // (workers * 8))
        with mp.get_context('spawn').Pool(workers, initializer=_front_init, initargs=(state, known, update_dict)) as pool:
            for label, docs, counts, new_words in pool.imap(_front_file, data_files, chunksize=chunk):
                for w in new_words:                      # first report in file order = the serial insertion order
                    if w not in self.dictionary:
                        self.dictionary[w] = 0
                for w, c in counts.items():
                    self.dictionary[w] += c
                if label:
                    label_dressed.update(label)
                    doc_dressed.extend(docs)
        return doc_dressed, label_dressed

    def _pool_size(self, n_files):
        """worker processes for load_data: params.load_workers (0 / 1 = serial); by default one per host core (at most
        16) once there are 64 files or more - never from a process that has already initialised the GPU (its children
        would be spawned from a GPU process)."""
        want = self.load_workers
        import sys
        main = sys.modules.get('__main__')
        if getattr(main, '__file__', None) is None and getattr(main, '__spec__', None) is None:
            return 1        # stdin / interactive: spawned workers cannot re-import __main__ (the pool would respawn them for ever)
        if want is None:
            if n_files < 64:
                return 1
            try:
                import torch
                if torch.cuda.is_initialized():
                    return 1
            except ImportError:
                pass
            want = min(16, os.cpu_count() or 1)
        return max(1, min(int(want), n_files))

    def _cal_rows_cols(self, docs, extra_augmentation=False, dropout=False):
        """:660-694"""
        max_row = max_col = self.encoding_factor
        for doc in docs:
            for line in doc:
                max_row = max(max_row, line[5])
                max_col = max(max_col, line[6])
        pre_rows = self._fit_shape(max_row)
        pre_cols = self._fit_shape(max_col)
        if not extra_augmentation:
            return pre_rows, pre_cols, pre_rows, pre_cols
        pad_row = int(random.gauss(0, self.da_extra_rows * self.encoding_factor))
        pad_col = int(random.gauss(0, self.da_extra_cols * self.encoding_factor))
        if self.augment_strategy == 1:
            rows = self._fit_shape(max_row + abs(pad_row))
            cols = self._fit_shape(max_col + abs(pad_col))
        elif self.augment_strategy in (2, 3):
            rows = self._fit_shape(max(self.rows_target + pad_row, max_row))
            cols = self._fit_shape(max(self.cols_target + pad_col, max_col))
        else:
            raise Exception('unknown augment strategy')
        return min(rows, self.rows_ulimit), min(cols, self.cols_ulimit), pre_rows, pre_cols

    def _fit_shape(self, shape):
        ef = self.encoding_factor
        return shape + (-shape) % ef

    def _expand_shape(self, shape):
        return self._fit_shape(shape + 1)

    def _natural_cols(self, boxes, extent, max_rows, max_cols):
        """Smallest column count at which every text line fits (:746-785): replay the shift-left
        placement on an occupancy board and widen by one whenever a row is completely full."""
        left, top, right, bottom = extent
        while True:
            board = [bytearray(max_cols) for _ in range(max_rows)]
            grown = False
            for (x_left, y_top, x_right, y_bottom) in boxes:
                # operator order of :754-755: (n * centre) / extent, not n * (centre / extent)
                row = int(max_rows * (y_top - top + (y_bottom - y_top) / 2) / (bottom - top))
                col = int(max_cols * (x_left - left + (x_right - x_left) / 2) / (right - left))
                cells = board[row]
                while col < max_cols and cells[col]:
                    col += 1
                if col == max_cols:
                    col -= 1
                    ptr = cells.find(0)
                    if ptr < 0:
                        max_cols = self._expand_shape(max_cols)
                        grown = True
                        break
                    del cells[ptr]          # shift the tail left by one ...
                    cells.append(1)         # ... the last cell keeps its (occupied) value
                cells[col] = 1
            if not grown:
                return max_cols

    def _collect_data(self, file_name, content, update_dict):
        """:704-831: items [file_name, token, word_id, [xl,yt,xr,yb], [left,top,right,bottom],
        max_rows, max_cols] per document (two lists when segment_grid splits a wide one)."""
        tokens = []
        left, top, right, bottom, buffer = 9999, 9999, 0, 0, 2
        for line in content:
            bbox = line['bbox']
            if len(bbox) == 0 or line['text'] in self.special_dict:
                continue
            x_left, y_top, x_right, y_bottom = self._dress_bbox(bbox)
            # compared against the already buffered extent, as the reference does (:719-722)
            if x_left < left:
                left = x_left - buffer
            if y_top < top:
                top = y_top - buffer
            if x_right > right:
                right = x_right + buffer
            if y_bottom > bottom:
                bottom = y_bottom + buffer
            pieces = self._dress_text(line['text'], update_dict)
            n = len(pieces)
            for i, piece in enumerate(pieces):      # a split word shares its box horizontally (:728-731)
                new_left = int(x_left + (x_right - x_left) / n * i)
                new_right = int(x_left + (x_right - x_left) / n * (i + 1))
                tokens.append([file_name, piece, line['id'], [new_left, y_top, new_right, y_bottom]])

        # most boxes crossing any one pixel row (:734-742): difference array + running sum
        edge = np.zeros(bottom + 1, dtype=np.int64)
        for t in tokens:
            _, y_top, _, y_bottom = t[3]
            if y_bottom > y_top:
                edge[y_top] += 1
                edge[y_bottom] -= 1
        max_row_words = self._fit_shape(int(np.cumsum(edge)[:bottom].max()))
        max_rows = max(self.encoding_factor, max_row_words)
        max_cols = self._natural_cols([t[3] for t in tokens], (left, top, right, bottom), max_rows,
                                      max(self.encoding_factor, 0))
        max_rows = self._fit_shape(max_rows)
        max_cols = self._fit_shape(max_cols)

        extent = [left, top, right, bottom]
        if not (self.segment_grid and max_cols > self.cols_segment):
            return [[t + [extent, max_rows, max_cols] for t in tokens]]

        # wide documents: left / right halves with cols_segment columns each (:789-824)
        part_l, part_r = [], []
        cnt, cnt_l, cnt_r = defaultdict(int), defaultdict(int), defaultdict(int)
        left_boundary = max_cols - self.cols_segment
        right_boundary = self.cols_segment
        for t in tokens:
            _, y_top, _, y_bottom = t[3]
            row = int(max_rows * (y_top + (y_bottom - y_top) / 2) / bottom)
            cnt[row] += 1
            # first left_boundary words of a row go left, the overlap band to both, the rest right
            to_l = cnt[row] <= left_boundary or cnt[row] <= right_boundary
            to_r = cnt[row] > left_boundary
            if to_l:
                cnt_l[row] += 1
                part_l.append(t + [extent, max_rows, self.cols_segment])
            if to_r:
                cnt_r[row] += 1
                part_r.append(t + [extent, max_rows, max(max(cnt_r.values()), self.cols_segment)])
        data = []
        if max(cnt_l.values()) < 2 * self.cols_segment:
            data.append(part_l)
        if max(cnt_r.values()) < 2 * self.cols_segment:
            data.append(part_r)
        return data

    def _collect_label(self, file_id, content):
        """:833-860 -> {file_id: {class: [{'key_id','value_id','key_text','value_text'}]}} or None"""
        per_class = {cls: [] for cls in self.classes[1:]}
        for line in content:
            cls = line['field_name']
            if cls in self.classes:
                per_class[cls].append({'key_id': line.get('key_id', []), 'value_id': line['value_id'],
                                       'key_text': line.get('key_text', []), 'value_text': line['value_text']})
        for entries in per_class.values():
            for label in entries:
                if (len(label['key_text']) > 0 and len(label['key_id']) == 0) or \
                   (len(label['value_text']) > 0 and len(label['value_id']) == 0):
                    return None
        return {file_id: per_class}

    def _dress_class(self, file_name, word_id, labels):
        """:862-885 -> (entity index within its class, class id); first match in dict order"""
        if file_name not in labels:
            print("No matched labels found for {}".format(file_name))
            return None
        for cls, cls_labels in labels[file_name].items():
            for idx, cls_label in enumerate(cls_labels):
                for key, values in cls_label.items():
                    if key == 'key_id' and word_id in values:
                        if self.data_mode == 0:
                            return idx, self.classes.index(cls) * 2 - 1
                        if self.data_mode == 1:
                            return idx, self.classes.index(cls)
                        return 0, 0
                    if key == 'value_id' and word_id in values:
                        if self.data_mode == 0:
                            return idx, self.classes.index(cls) * 2
                        return idx, self.classes.index(cls)
        return 0, 0

    def _dress_text(self, text, update_dict):
        """:887-925: optional lower-casing, every numeric character -> '0', WordPiece split,
        unknown pieces -> '[UNK]' (or new dictionary entries when update_dict); counts words"""
        string = text if self.text_case else text.lower()
        # the reference rebuilds the string while enumerating the ORIGINAL one; a replacement
        # never changes the length, so a per-character map is the same thing
        string = ''.join('0' if is_number(c) else c for c in string)
        strings = self.tokenizer.tokenize(string) if self.tokenize else [string]
        for idx, s in enumerate(strings):
            if s not in self.dictionary:
                if update_dict:
                    self.dictionary[s] = 0
                else:
                    s = '[UNK]'
            self.dictionary[s] += 1
            strings[idx] = s
        return strings

    def _dress_bbox(self, bbox):
        """:927-935: any polygon -> (x_left, y_top, x_right, y_bottom), left/top clamped at 0"""
        pos = np.array(bbox).reshape([-1])
        xs, ys = pos[0::2], pos[1::2]
        return int(max(0, min(xs))), int(max(0, min(ys))), int(max(xs)), int(max(ys))

    def _get_filenames(self, data_path):
        """:937-944"""
        files = []
        for dirpath, _dirnames, filenames in os.walk(data_path):
            for filename in filenames:
                if filename.endswith('csv') or filename.endswith('json'):
                    files.append(os.path.join(dirpath, filename))
        return files
