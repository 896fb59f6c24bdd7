"""Names and sampling rates the path depends on (values of the reference's `settings.py`; the dataset names are the ones its
validation loop keys on, `trainer/main.py:188-224`)."""

# signal columns, grouped by samples per 30-second epoch (respiratory belts 256, cardiac 1024, EOG 4096)
_SIGNALS_BY_RATE = {256: ('ABD', 'THX'), 1024: ('ECG', 'PPG'), 4096: ('EOG-L', 'EOG-R')}
COLS_TO_SAMPLES_PER_EPOCH = {name: rate for rate, names in _SIGNALS_BY_RATE.items() for name in names}
ABD, THX, ECG, PPG, EOG_L, EOG_R = (n for names in _SIGNALS_BY_RATE.values() for n in names)
LOW_FREQ_SAMPLES_PER_EPOCH, MEDIUM_FREQ_SAMPLES_PER_EPOCH, HIGH_FREQ_SAMPLES_PER_EPOCH = sorted(_SIGNALS_BY_RATE)

# parquet / csv column names
LABEL, TIMESTAMP, PRED = 'Stage', 'Timestamp', 'Pred'

# causal (online) normalisation of the dataset, `ParquetDataset(causal=True)`: seconds / sigmas
CAUSAL_NORM_TAU_SECONDS = 900.0            # variance tracking
CAUSAL_NORM_BASELINE_TAU_SECONDS = 120.0   # baseline (mean) tracking
NORM_OUTLIER_THRESHOLD = 4.0               # residuals beyond this many sigma are clipped before they enter the variance
CAUSAL_NORM_MIN_SIGMA = 0.1                # sigma floor

TRAINING_LENGTH_HOURS = 10   # recordings are padded / cropped to this length for training
TRAIN, VAL, TEST = 'train', 'val', 'test'

# five annotated stages (W, N1, N2, N3, REM) -> class index; the 4-class problem merges N1 and N2 into "light"
INTEGER_LABEL_MAPS = {
    5: {stage: stage for stage in range(5)},
    4: dict(zip(range(5), (0, 1, 1, 2, 3))),
}
