// aisx_framing.cpp -- host-side tail of the receive chain:
// digital.hdlc_deframer_bp(length_min, length_max) as python/radio.py:64 uses it
// ([GR] gr-digital hdlc_deframer_bp_impl::work: flag 0x7E search, bit unstuffing,
// bytes packed LSB first, CRC-16/X.25) and ais.pdu_to_nmea
// (lib/pdu_to_nmea_impl.cc:63-131: 6-bit unpack, ASCII armouring, 56-character
// fragments, XOR checksum).  Per-packet, bytes-per-second work: plain host C++,
// no GPU involved.
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/aisx.h"

struct aisx_hdlc {
    int length_min, length_max;
    std::vector<unsigned char> pktbuf;
    int ones = 0, bitctr = 0, bytectr = 0;
};

static unsigned short crc_ccitt(const unsigned char* data, size_t len)
{
    const unsigned POLY = 0x8408; // reflected 0x1021
    unsigned short crc = 0xFFFF;
    for (size_t i = 0; i < len; i++) {
        crc ^= data[i];
        for (int j = 0; j < 8; j++)
            crc = (crc & 0x01) ? (unsigned short)((crc >> 1) ^ POLY) : (unsigned short)(crc >> 1);
    }
    return crc ^ 0xFFFF;
}

extern "C" int aisx_hdlc_create(aisx_hdlc** h, int length_min, int length_max)
{
    if (!h || length_min < 0 || length_max < length_min)
        return AISX_ERR_INVALID;
    aisx_hdlc* d = new aisx_hdlc();
    d->length_min = length_min;
    d->length_max = length_max;
    d->pktbuf.assign(length_max + 2, 0);
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
    int np = 0, used = 0, rc = AISX_OK;
    if (max_pdus > 0)
        pdu_offsets[0] = 0;
    for (int i = 0; i < nbits; i++) {
        const unsigned char bit = bits[i];
        if (h->ones >= 5) {
            if (bit) { // six ones: frame delimiter
                if (h->bytectr >= h->length_min) {
                    const int len = h->bytectr - 2;
                    const unsigned short crc = crc_ccitt(h->pktbuf.data(), len);
                    const unsigned short pktcrc = (unsigned short)(h->pktbuf[len + 1] << 8 | h->pktbuf[len]);
                    if (crc == pktcrc) {
                        if (np < max_pdus && used + len <= pdu_cap) {
                            memcpy(pdu_bytes + used, h->pktbuf.data(), len);
                            used += len;
                            pdu_offsets[np + 1] = used;
                        } else {
                            rc = AISX_ERR_OVERFLOW;
                        }
                        np++;
                    }
                    h->pktbuf.assign(h->length_max + 2, 0);
                }
                h->bitctr = 0;
                h->bytectr = 0;
            } // else: a stuffed zero, dropped
        } else {
            if (h->bytectr > h->length_max) { // overran the packet buffer
                h->bitctr = 0;
                h->bytectr = 0;
                h->pktbuf.assign(h->length_max + 2, 0);
            } else {
                h->pktbuf[h->bytectr] >>= 1;
                if (bit)
                    h->pktbuf[h->bytectr] |= 0x80;
                h->bitctr++;
                if (h->bitctr == 8) {
                    h->bitctr = 0;
                    h->bytectr++;
                }
            }
        }
        h->ones = bit ? h->ones + 1 : 0;
    }
    *npdus = np;
    return rc;
}

// lib/pdu_to_nmea_impl.cc:63-131
extern "C" int aisx_pdu_to_nmea(const char* designator, const uint8_t* pdu, int len, char* out, int cap)
{
    if (!designator || !pdu || len < 1 || !out || cap < 1)
        return AISX_ERR_INVALID;
    // unpack_bits (:63-79)
    const int nbits = len * 8;
    const int npad = (6 - (nbits % 6)) % 6;
    std::vector<unsigned char> up((nbits + npad) / 6, 0);
    for (int i = 0; i < nbits; i++) {
        const unsigned char bit = (pdu[i / 8] >> (7 - (i % 8))) & 1;
        up[i / 6] |= (unsigned char)(bit << (5 - (i % 6)));
    }
    for (int i = 0; i < npad; i++)
        up[nbits / 6] <<= 1;
    // to_ascii (:81-88)
    std::string ascii(up.begin(), up.end());
    for (size_t i = 0; i < ascii.size(); i++) {
        if (ascii[i] > 39)
            ascii[i] += 8;
        ascii[i] += char(48);
    }
    // to_sentence (:99-124)
    const int nmea_max = 56;
    const int num_frags = 1 + (((int)ascii.length() - 1) / nmea_max);
    std::string ret;
    int frag_id = 1;
    size_t frag_offset = 0;
    while (frag_id <= num_frags) {
        if (frag_id > 1)
            ret += "\n";
        std::string s = "!AIVDM," + std::to_string(num_frags) + "," + std::to_string(frag_id++) + ",," + designator + ",";
        std::string frag = ascii.substr(frag_offset, nmea_max);
        frag_offset += frag.length();
        s += frag + "," + std::to_string(npad);
        unsigned char sum = 0; // get_checksum (:90-96)
        for (size_t i = (s[0] == '!') ? 1 : 0; i < s.length(); i++)
            sum ^= (unsigned char)s[i];
        char wat[3];
        snprintf(wat, 3, "%02X", sum);
        s += "*" + std::string(wat);
        ret += s;
    }
    if ((int)ret.size() + 1 > cap)
        return AISX_ERR_OVERFLOW;
    memcpy(out, ret.c_str(), ret.size() + 1);
    return (int)ret.size();
}
