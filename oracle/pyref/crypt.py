"""Second, independent restatement (plain Python, written from the C#) of the ADX / HCA encryption code:
VGAudio/Codecs/CriAdx/CriAdxKey.cs:10-66, CriAdxEncryption.cs:8-108, Utilities/Helpers.cs:115-139 (GetPrimes),
VGAudio/Codecs/CriHca/CriHcaKey.cs:8-181, CriHcaEncryption.cs:12-88.  Test infrastructure only: it pins oracle/crypt_oracle.c
(tests/test_pyref_crosscheck.py)."""
from . import crihca as _hca


def get_primes(max_prime):                                        # Helpers.cs:115-139
    mx = max_prime // 2
    sieve = bytearray(mx)
    i = 3
    while i * i < max_prime:
        if sieve[i >> 1] == 0:
            for j in range(i * i, max_prime, i * 2):
                sieve[j >> 1] = 1
        i += 2
    primes = [2]
    for i in range(1, mx):
        if sieve[i] == 0:
            primes.append(i * 2 + 1)
    return primes


def _build_primes_table():                                        # CriAdxKey.cs:58-65
    primes = get_primes(0x8000)
    # ~Array.BinarySearch(primes, 0x4000): 0x4000 is not prime, so this is the index of the first prime above it
    start = next(k for k, v in enumerate(primes) if v > 0x4000)
    return primes[start:start + 0x400]


PRIMES = _build_primes_table()


class AdxKey:
    def __init__(self, seed=0, mult=0, inc=0):
        self.seed, self.mult, self.inc = seed, mult, inc

    @staticmethod
    def from_code(key_code):                                      # CriAdxKey.cs:17-23
        k = (key_code - 1) & 0xFFFFFFFFFFFFFFFF
        return AdxKey((k >> 27) & 0x7fff, ((k >> 12) & 0x7ffc) | 1, ((k << 1) & 0x7fff) | 1)

    @staticmethod
    def from_string(s):                                           # :25-39
        if not s:
            return AdxKey()
        seed, mult, inc = PRIMES[0x100], PRIMES[0x200], PRIMES[0x300]
        for ch in s:
            c = ord(ch)
            seed = PRIMES[seed * PRIMES[c + 0x80] % 0x400]
            mult = PRIMES[mult * PRIMES[c + 0x80] % 0x400]
            inc = PRIMES[inc * PRIMES[c + 0x80] % 0x400]
        return AdxKey(seed, mult, inc)

    def key_code(self):                                           # :47-56
        return ((((self.seed & 0xFFFFFFFFFFFFFFFF) << 27) | ((self.mult & 0xfffc) << 12) | ((self.inc & 0xFFFFFFFFFFFFFFFF) >> 1)) + 1) \
            & 0xFFFFFFFFFFFFFFFF


def _i32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def adx_crypt_channel(adpcm, key, encryption_type, frame_size, channel_num, channel_count):   # CriAdxEncryption.cs:16-41
    """in place on a bytearray"""
    xor = key.seed
    frame_count = (len(adpcm) + frame_size - 1) // frame_size
    for _ in range(channel_num):
        xor = _i32(xor * key.mult + key.inc) & 0x7fff
    for i in range(frame_count):
        pos = i * frame_size
        if any(adpcm[pos:pos + frame_size]):                      # FrameNotEmpty (a short last frame would throw in C#)
            adpcm[pos] ^= (xor >> 8) & 0xff
            if encryption_type == 9:
                adpcm[pos] &= 0x1f
            adpcm[pos + 1] ^= xor & 0xff
        for _ in range(channel_count):
            xor = _i32(xor * key.mult + key.inc) & 0x7fff


def adx_test_key(channels, key, encryption_type, frame_size):     # GetScales + TestKey, :58-93
    frame_count = (len(channels[0]) + frame_size - 1) // frame_size
    mask = 0xE000 if encryption_type == 8 else 0x1000
    xor = key.seed
    for frame in range(frame_count):
        pos = frame * frame_size
        for ch in channels:
            scale = (ch[pos] << 8) | ch[pos + 1]
            if ((scale ^ xor) & mask) != 0 and scale != 0:
                return False
            xor = _i32(xor * key.mult + key.inc) & 0x7fff
    return True


# ---------------------------------------------------------------- HCA
def _random_row(seed):                                            # CriHcaKey.cs:118-131
    xor = seed >> 4
    mult = ((seed & 1) << 3) | 5
    inc = (seed & 0xe) | 1
    row = []
    for _ in range(16):
        xor = (xor * mult + inc) % 16
        row.append(xor)
    return row


ROWS = [_random_row(i) for i in range(256)]


def _shuffle(table_in):                                           # :145-161
    table = [0] * 256
    x, out_pos = 0, 1
    for _ in range(256):
        x = (x + 17) & 0xff
        if table_in[x] != 0 and table_in[x] != 0xff:
            table[out_pos] = table_in[x]
            out_pos += 1
    table[0xff] = 0xff
    return table


def _create_table(row_seed, column_seeds):                        # :101-116
    table = [0] * 256
    row = ROWS[row_seed]
    for r in range(16):
        column = ROWS[column_seeds[r]]
        for c in range(16):
            table[16 * r + c] = ((row[r] << 4) | column[c]) & 0xff     # Helpers.CombineNibbles
    return _shuffle(table)


def hca_decryption_table(key_type, key_code=0):                   # :8-99
    if key_type == 0:
        return list(range(256))
    if key_type == 1:
        table = [0] * 256
        xor, out_pos = 0, 1
        for _ in range(256):
            xor = (xor * 13 + 11) % 256
            if xor != 0 and xor != 0xff:
                table[out_pos] = xor
                out_pos += 1
        table[0xff] = 0xff
        return table
    kc = list(((key_code - 1) & 0xFFFFFFFFFFFFFFFF).to_bytes(8, "little"))
    seed = [kc[1], kc[6] ^ kc[1], kc[2] ^ kc[3], kc[2], kc[1] ^ kc[2], kc[3] ^ kc[4], kc[3], kc[2] ^ kc[3],
            kc[4] ^ kc[5], kc[4], kc[3] ^ kc[4], kc[5] ^ kc[6], kc[5], kc[4] ^ kc[5], kc[6] ^ kc[1], kc[6]]
    return _create_table(kc[0], seed)


def invert_table(table):                                          # :163-174
    out = [0] * len(table)
    for i, v in enumerate(table):
        out[v] = i
    return out


def hca_crypt_frame(frame, frame_size, table):                    # CriHcaEncryption.cs:20-32; in place on a bytearray
    for b in range(frame_size - 2):
        frame[b] = table[frame[b]]
    crc = _hca.crc16(frame, frame_size - 2)
    frame[frame_size - 2] = (crc >> 8) & 0xff
    frame[frame_size - 1] = crc & 0xff


def hca_find_key(info, frames, decryption_tables):                # :34-88 over caller-supplied keys; index or -1
    """frames: list of bytes-like, info: crihca.HcaInfo-like object the unpacker accepts."""
    def frame_empty(f):
        return not any(f[2:len(f) - 2])
    start = next((i for i, f in enumerate(frames) if not frame_empty(f)), 0)
    end = min(len(frames), start + 10)
    frame = _hca.Frame(info)                                      # one CriHcaFrame for every key and frame, as :36
    for k, table in enumerate(decryption_tables):
        ok = True
        for i in range(start, end):
            buf = bytearray(frames[i])
            hca_crypt_frame(buf, info.frame_size, table)
            if not _hca._unpack_frame(frame, _hca.BitReader(bytes(buf))):   # ValueError: invalid sync word (InvalidDataException)
                ok = False
                break
        if ok:
            return k
    return -1


# ---------------------------------------------------------------- VGAudio.Tools/CrackAdx/GuessAdx.cs (one file)
def adx_file(audio, frame_size):                                  # AdxFile :268-297: scales and the first non-empty frame
    frame_count = len(audio) // frame_size
    scales = [(audio[i * frame_size] << 8) | audio[i * frame_size + 1] for i in range(frame_count)]
    start = 0
    for i, b in enumerate(audio):
        if b != 0:
            start = i // frame_size
            break
    return scales, start


def adx_guess_candidates(encryption_type):                        # GuessAdx constructor :50-72
    if encryption_type == 8:
        return list(PRIMES), list(PRIMES), set(PRIMES), 0xE000, 0x8000
    return ([x for x in range(0x2000) if (x & 3) == 1], [x for x in range(0x2000) if (x & 1) == 1], set(range(0x2000)),
            0x1000, 0x2000)


def adx_guess_keys(scales, start_frame, encryption_type, mults=None, incs=None):
    """Run / TryScale / FindStartingKey / AddKey's KeyIsValid for ONE file (:113-218): the keys every scale agrees with,
    without duplicates, sorted by (seed, mult, inc)."""
    dm, di, seeds, vmask, max_seed = adx_guess_candidates(encryption_type)
    mults = dm if mults is None else list(mults)
    incs = di if incs is None else list(incs)
    xmask = 0x7fff
    found = set()

    def key_is_valid(seed, mult, inc):                            # :208-218
        xor = seed
        for scale in scales:
            if ((scale ^ xor) & vmask) != 0 and scale != 0:
                return False
            xor = _i32(xor * mult + inc) & xmask
        return True

    for index in range(0x1000):                                   # Run :113-124
        seed = (scales[start_frame] ^ index) & (max_seed - 1)     # TryScale :160-163
        if start_frame == 0 and seed not in seeds:
            continue
        for mult in mults:
            for inc in incs:
                xor, match = seed, True
                for i in range(start_frame, len(scales)):
                    scale = scales[i]
                    if ((scale ^ xor) & vmask) != 0 and scale != 0:
                        match = False
                        break
                    xor = _i32(xor * mult + inc) & xmask
                if not match:
                    continue
                key = None                                        # FindStartingKey :187-206
                if start_frame == 0:
                    key = (seed, mult, inc)
                else:
                    for real_seed in sorted(seeds):               # HashSet order of ints inserted ascending: ascending
                        xor = real_seed
                        for _ in range(start_frame):
                            xor = _i32(xor * mult + inc) & xmask
                        if (xor & (max_seed - 1)) == seed:
                            key = (real_seed, mult, inc)
                            break
                if key is not None and key not in found and key_is_valid(*key):   # AddKey :128-135
                    found.add(key)
    return sorted(found)
