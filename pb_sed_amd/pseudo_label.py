"""Pseudo-labelling hand-off: what the ensemble inference of one round writes for the training of the next.

Host-side counterpart of ``pb_sed/models/base/pseudo_label.py`` (reference: ``pseudo_label`` :5-47, ``set_onset_offset_times``
:50-69; called at pb_sed/experiments/strong_label_crnn/inference.py:387-400 and
pb_sed/experiments/weak_label_crnn/inference.py with the tags / boundaries / event lists that
``pb_sed_amd.inference`` produces on the GPU).  Plain dictionary work - no device code; pinned by
``tests/golden/ref_pseudo_label.json``, which the reference's own functions produced (tests/golden/gen_golden.py).

A dataset is ``{audio_id: example}``; an example carries ``events`` (class names) and ``audio_length`` (seconds), after this
step also ``label_types`` and - where detections were handed in - ``events_start_times`` / ``events_stop_times``.
"""
import copy

LABEL_TYPES = ('weak', 'boundaries', 'strong')


def set_onset_offset_times(example, detections, label_type='strong'):
    """Give ``example`` the on-/offsets of ``detections`` [(onset, offset, label)] for the classes it is tagged with.  A tagged
    class without a detection keeps a weak label over the whole clip; detections of classes the clip is not tagged with are
    dropped.  Writes ``events_start_times`` / ``events_stop_times`` / ``events`` (sorted by (onset, offset, label)) and
    ``label_types`` in place."""
    assert 'events' in example, sorted(example)
    tagged = sorted(set(example['events']))
    detected = {label for _, _, label in detections}
    rows = [tuple(d) for d in detections if d[2] in tagged]
    rows += [(0., example['audio_length'], label) for label in tagged if label not in detected]
    rows.sort()
    if rows:
        starts, stops, labels = (tuple(col) for col in zip(*rows))
    else:
        starts, stops, labels = [], [], []
    example['events_start_times'], example['events_stop_times'], example['events'] = starts, stops, labels
    example['label_types'] = [label_type if label in detected else 'weak' for label in labels]
    return example


def pseudo_label(dataset, event_classes, pseudo_tags, pseudo_boundaries, pseudo_events, tags, boundaries, events, verbose=False):
    """A deep copy of ``dataset`` relabelled with the ensemble's output; ``dataset`` itself when nothing is asked for.

    ``pseudo_tags``: ``events`` of every clip become the classes whose tag score ``tags[audio_id][k]`` exceeds 0.5
    (``event_classes[k]``); ``pseudo_events`` / ``pseudo_boundaries`` (mutually exclusive): on-/offsets from ``events[audio_id]`` /
    ``boundaries[audio_id]`` through ``set_onset_offset_times`` with label type 'strong' / 'boundaries'.  Every clip gets
    ``label_types`` ('weak' for all its events unless detections replace them).  ``verbose`` prints the label rates the
    reference prints."""
    if not (pseudo_tags or pseudo_boundaries or pseudo_events):
        return dataset
    assert not (pseudo_events and pseudo_boundaries), 'strong and boundary pseudo labels exclude each other'
    out = copy.deepcopy(dataset)
    for audio_id in sorted(out):
        example = out[audio_id]
        if pseudo_tags:
            example['events'] = sorted(name for score, name in zip(tags[audio_id], event_classes) if score > 0.5)
        example['label_types'] = ['weak'] * len(example['events'])
        if pseudo_events:
            set_onset_offset_times(example, events[audio_id], 'strong')
        elif pseudo_boundaries:
            set_onset_offset_times(example, boundaries[audio_id], 'boundaries')
    if verbose:
        rates = label_rates(out)
        print(f"\nlabel rate {rates['label_rate']}")
        for kind in LABEL_TYPES:
            print(f'pseudo {kind} labels rate {rates[kind]}')
    return out


def label_rates(dataset):
    """The figures the reference prints after relabelling: share of clips with at least one event, and the share of each label
    type among all events."""
    ids = sorted(dataset)
    kinds = [t for a in ids for t in dataset[a]['label_types']]
    out = {'label_rate': sum(len(dataset[a]['events']) > 0 for a in ids) / len(ids) if ids else float('nan')}
    for kind in LABEL_TYPES:
        out[kind] = sum(t == kind for t in kinds) / len(kinds) if kinds else float('nan')
    return out


def write_event_tsv(path, events):
    """``<dataset>_pseudo_labeled.tsv`` as the reference's inference script writes it
    (pb_sed/experiments/strong_label_crnn/inference.py:393-400): one row per detected event, an empty row for a clip without."""
    with open(path, 'w') as f:
        f.write('filename\tonset\toffset\tevent_label\n')
        for key, event_list in events.items():
            if len(event_list) == 0:
                f.write(f'{key}.wav\t\t\t\n')
            for onset, offset, label in event_list:
                f.write(f'{key}.wav\t{onset}\t{offset}\t{label}\n')
