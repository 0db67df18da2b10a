// aisx_framing.cpp -- host-side tail of the receive chain, plain C++:
//
//   aisx_hdlc_*      what digital.hdlc_deframer_bp(length_min, length_max) does for
//                    python/radio.py:64: HDLC bit de-stuffing, frames delimited by six
//                    consecutive ones, octets filled LSB first, CRC-16/X.25 over all but the
//                    last two octets, which carry the FCS low byte first;
//   aisx_pdu_to_nmea the sentence format of ais.pdu_to_nmea (observable behaviour of
//                    lib/pdu_to_nmea_impl.cc:63-131): ITU-R M.1371 / NMEA 0183 six-bit
//                    armouring, 56 payload characters per !AIVDM fragment, XOR checksum.
//
// Per-packet, bytes-per-second work: no GPU involved.  Written from the protocol rules; the
// reference's observable quirks that a consumer could depend on are kept and named below.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/aisx.h"

namespace {

// CRC-16/X.25: polynomial 0x1021 reflected (0x8408), preset 0xFFFF, complemented result
class X25Fcs {
public:
    X25Fcs()
    {
        for (unsigned v = 0; v < 256; v++) {
            unsigned r = v;
            for (int k = 0; k < 8; k++)
                r = (r >> 1) ^ ((r & 1u) ? 0x8408u : 0u);
            table_[v] = (uint16_t)r;
        }
    }
    uint16_t operator()(const uint8_t* octets, size_t count) const
    {
        unsigned reg = 0xFFFFu;
        for (size_t k = 0; k < count; k++)
            reg = (reg >> 8) ^ table_[(reg ^ octets[k]) & 0xFFu];
        return (uint16_t)(~reg & 0xFFFFu);
    }

private:
    uint16_t table_[256];
};

const X25Fcs& fcs()
{
    static const X25Fcs f;
    return f;
}

} // namespace

// Receiver state: the run of ones seen so far (the de-stuffer), the octet being filled and the
// octets of the frame under way.
struct aisx_hdlc {
    int min_octets = 0, max_octets = 0;
    unsigned ones_run = 0;
    unsigned shift = 0;   // octet under construction, filled from the top and shifted down
    int nshift = 0;       // bits in it
    std::vector<uint8_t> frame;

    void drop_frame()
    {
        frame.clear();
        shift = 0;
        nshift = 0;
    }
    void data_bit(unsigned bit)
    {
        // a frame that has outgrown length_max is abandoned as soon as one more data bit shows
        // up (the opening bits of the closing flag count: a frame of length_max + 1 octets never
        // survives, one of exactly length_max does)
        if ((int)frame.size() > max_octets) {
            drop_frame();
            return;
        }
        shift = (shift >> 1) | (bit ? 0x80u : 0u); // first bit received ends up as bit 0
        if (++nshift == 8) {
            frame.push_back((uint8_t)shift);
            shift = 0;
            nshift = 0;
        }
    }
    // six ones in a row: end of frame (or an abort / idle flags when nothing was collected).
    // Whole octets only; the flag's own leading bits sit in the partial octet and go with it.
    template <class Sink>
    void delimiter(Sink&& deliver)
    {
        const int got = (int)frame.size();
        if (got >= min_octets) {
            const int payload = got - 2;
            const unsigned sent = (unsigned)frame[payload] | ((unsigned)frame[payload + 1] << 8);
            if (fcs()(frame.data(), (size_t)payload) == sent)
                deliver(frame.data(), payload);
        }
        drop_frame();
    }
};

extern "C" int aisx_hdlc_create(aisx_hdlc** h, int length_min, int length_max)
{
    // a frame is its payload plus the two FCS octets: anything shorter has no payload to check
    if (!h || length_min < 2 || length_max < length_min)
        return AISX_ERR_INVALID;
    aisx_hdlc* d = new aisx_hdlc();
    d->min_octets = length_min;
    d->max_octets = length_max;
    d->frame.reserve((size_t)length_max + 2);
    *h = d;
    return AISX_OK;
}

extern "C" int aisx_hdlc_destroy(aisx_hdlc* h)
{
    delete h;
    return AISX_OK;
}

extern "C" int aisx_hdlc_work(aisx_hdlc* h, const uint8_t* bits, int nbits, uint8_t* pdu_bytes, int pdu_cap,
                              int* pdu_offsets, int max_pdus, int* npdus)
{
    if (!h || !bits || nbits < 0 || !npdus || (max_pdus > 0 && (!pdu_offsets || !pdu_bytes)))
        return AISX_ERR_INVALID;
    int found = 0, fill = 0, status = AISX_OK;
    if (max_pdus > 0)
        pdu_offsets[0] = 0;
    auto deliver = [&](const uint8_t* octets, int count) {
        if (found < max_pdus && fill + count <= pdu_cap) {
            memcpy(pdu_bytes + fill, octets, (size_t)count);
            fill += count;
            pdu_offsets[found + 1] = fill;
        } else {
            status = AISX_ERR_OVERFLOW;
        }
        found++;
    };
    for (const uint8_t *b = bits, *end = bits + nbits; b != end; ++b) {
        const unsigned bit = *b ? 1u : 0u;
        if (h->ones_run < 5)
            h->data_bit(bit);
        else if (bit)
            h->delimiter(deliver);
        // (else: the zero the transmitter stuffed behind five ones -- not data)
        h->ones_run = bit ? h->ones_run + 1 : 0;
    }
    *npdus = found;
    return status;
}

namespace {

// one payload character from a six-bit value: 0..39 -> '0'..'W', 40..63 -> '`'..'w'.
// Quirk kept from the reference (lib/pdu_to_nmea_impl.cc:81-88 compares a plain `char`): the
// padded last group can exceed 63, and a value of 128 or more counts as negative there.
inline char armour(unsigned group)
{
    const int as_char = (int)(int8_t)(uint8_t)group;
    return (char)(as_char + (as_char > 39 ? 56 : 48));
}

// bounded text sink that keeps the NMEA checksum (XOR of everything between '!' and '*')
class SentenceWriter {
public:
    SentenceWriter(char* dst, int cap) : dst_(dst), cap_(cap) {}
    void raw(char c)
    {
        if (used_ < cap_)
            dst_[used_] = c;
        used_++;
    }
    void put(char c)
    {
        sum_ ^= (uint8_t)c;
        raw(c);
    }
    void put(const char* s)
    {
        while (*s)
            put(*s++);
    }
    void put_number(int v)
    {
        char tmp[16];
        snprintf(tmp, sizeof tmp, "%d", v);
        put(tmp);
    }
    void open()
    {
        raw('!');
        sum_ = 0;
    }
    void close()
    {
        static const char hex[] = "0123456789ABCDEF";
        const uint8_t s = sum_;
        raw('*');
        raw(hex[s >> 4]);
        raw(hex[s & 15]);
    }
    int used() const { return used_; }

private:
    char* dst_;
    int cap_, used_ = 0;
    uint8_t sum_ = 0;
};

} // namespace

extern "C" int aisx_pdu_to_nmea(const char* designator, const uint8_t* pdu, int len, char* out, int cap)
{
    if (!designator || !pdu || len < 1 || !out || cap < 1)
        return AISX_ERR_INVALID;
    // payload: the PDU's bits, most significant first, in groups of six
    const int fill_bits = (6 - (len * 8) % 6) % 6;
    std::vector<char> payload;
    payload.reserve((size_t)(len * 8 + fill_bits) / 6);
    unsigned window = 0;
    int held = 0;
    for (int k = 0; k < len; k++) {
        window = ((window << 8) | pdu[k]) & 0xFFFFu;
        held += 8;
        while (held >= 6) {
            held -= 6;
            payload.push_back(armour((window >> held) & 63u));
        }
    }
    if (held > 0) {
        // Quirk kept from the reference (lib/pdu_to_nmea_impl.cc:70-78): the left-over bits are
        // placed at the top of their group and the group is then shifted up by the fill count
        // once more, in eight bits -- with 4 fill bits the group always comes out as 0
        const unsigned top_aligned = (window & ((1u << held) - 1u)) << (6 - held);
        payload.push_back(armour((top_aligned << fill_bits) & 0xFFu));
    }
    // fragments of at most 56 payload characters; every fragment reports the fill count (:113)
    const int per_fragment = 56;
    const int total = (int)payload.size();
    const int fragments = (total + per_fragment - 1) / per_fragment;
    SentenceWriter w(out, cap - 1);
    for (int f = 0; f < fragments; f++) {
        if (f)
            w.raw('\n');
        w.open();
        w.put("AIVDM,");
        w.put_number(fragments);
        w.put(',');
        w.put_number(f + 1);
        w.put(",,");
        w.put(designator);
        w.put(',');
        const int first = f * per_fragment, last = first + per_fragment < total ? first + per_fragment : total;
        for (int k = first; k < last; k++)
            w.put(payload[(size_t)k]);
        w.put(',');
        w.put_number(fill_bits);
        w.close();
    }
    if (w.used() > cap - 1)
        return AISX_ERR_OVERFLOW;
    out[w.used()] = '\0';
    return w.used();
}
