// abi_fw.cpp -- part of the C-ABI of libsynth_mi355x.so (include/synth_mi355x.h): hosted firmware control surface (TAG_U32)
// Host side of the drop-in boundary.  No CPU compute fallback exists: every compute entry
// point needs a HIP device and fails with SMX_E_NOGPU otherwise.
#include "abi_internal.h"
// ---------------------------------------------------------------------------
// firmware control surface, hosted: mod_synth.c:50-137, stm32f103/synth.c:27-42
// ---------------------------------------------------------------------------
struct smx_fw {
    smx_pwm *pwm = nullptr;
    smx_osc *osc = nullptr;
    int running = 0;
    uint32_t param[1] = {15u << 27};            // osc_setpoint, mod_synth.c:50-51
    std::vector<uint32_t> read_idx;             // pmeas_state.read per oscillator
    std::vector<uint8_t> wait;                  // measurement_wait: {len, bytes...}*, 64 bytes (mod_osc.c:43)
};

extern "C" smx_fw *smx_fw_create(uint32_t n_channels, uint32_t n_osc, int device)
{
    smx_fw *f = new smx_fw();
    f->pwm = smx_pwm_create(n_channels, 2, device);          // PDM_ORDER 2
    f->osc = n_osc ? smx_osc_create(n_osc, device) : nullptr;
    if (!f->pwm || (n_osc && !f->osc) || smx_pwm_init(f->pwm) != SMX_OK) {   // pdm_init
        smx_fw_destroy(f);
        return nullptr;
    }
    f->running = 1;                                          // pdm_start, mod_synth.c:67
    f->read_idx.assign(n_osc, 0);
    return f;
}

extern "C" void smx_fw_destroy(smx_fw *f)
{
    if (!f) return;
    smx_pwm_destroy(f->pwm);
    smx_osc_destroy(f->osc);
    delete f;
}

extern "C" smx_pwm *smx_fw_pwm(smx_fw *f) { return f ? f->pwm : nullptr; }
extern "C" smx_osc *smx_fw_osc(smx_fw *f) { return f ? f->osc : nullptr; }
extern "C" int smx_fw_running(const smx_fw *f) { return f ? f->running : 0; }
extern "C" uint32_t smx_fw_parameter(const smx_fw *f, uint32_t id) { return (f && id < 1) ? f->param[id] : 0; }

extern "C" int smx_fw_handle_tag_u32(smx_fw *f, const uint32_t *args, uint32_t nb_args,
                                     const uint8_t *bytes, uint32_t nb_bytes)
{
    if (!f || (nb_args && !args) || (nb_bytes && !bytes)) return -1;
    if (nb_args < 1) return -1;                              // mod_synth.c:91
    switch (args[0]) {
    case 100:                                                // MODE, :97-103
        if (nb_args < 2) return -1;
        f->running = args[1] ? 1 : 0;
        return 0;
    case 101:                                                // SETPOINT, :104-111
        if (nb_args < 3) return -1;
        return smx_pwm_set_setpoint(f->pwm, args[1], args[2]);   // -2 when chan >= PDM_NB_CHANNELS
    case 102:                                                // MEASURE, :112-127
        if (nb_args > 2) return -3;
        if (nb_args == 2 && f->osc) {
            // the firmware stores any value; the shift in pmeas_update is only defined for 1..31
            if (smx_osc_set_log_max(f->osc, args[1]) != SMX_OK) return -1;
        }
        if (nb_bytes) {
            // cbuf_put(len); cbuf_write(bytes): a 64-byte ring in the firmware
            if (nb_bytes > 255 || f->wait.size() + 1 + nb_bytes > 64) return 0;   // no room: dropped
            f->wait.push_back((uint8_t)nb_bytes);
            f->wait.insert(f->wait.end(), bytes, bytes + nb_bytes);
        }
        return 0;
    default:                                                 // parameter table, :129-135
        if (nb_args < 2 || args[0] >= 1) return -1;
        f->param[args[0]] = args[1];
        return 0;
    }
}

static inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

extern "C" int smx_fw_handle_packet(smx_fw *f, const uint8_t *buf, uint32_t len)
{
    if (!f || !buf || len < 2) return -1;
    const uint32_t tag = ((uint32_t)buf[0] << 8) | buf[1];
    if (tag != SMX_TAG_U32) {
        fprintf(stderr, "unknown tag 0x%x\n", tag);          // stm32f103/synth.c:39-40
        return 0;
    }
    if (len < 4) return -1;
    const uint32_t nb_from = buf[2], nb_args = buf[3];
    const uint64_t words = (uint64_t)nb_from + nb_args;
    if (4 + 4 * words > len) return -1;
    std::vector<uint32_t> args(nb_args);
    for (uint32_t i = 0; i < nb_args; i++) args[i] = be32(buf + 4 + 4 * (nb_from + i));
    const uint32_t off = 4 + 4 * (uint32_t)words;
    const int rv = smx_fw_handle_tag_u32(f, args.data(), nb_args, buf + off, len - off);
    if (rv) fprintf(stderr, "tag_u32_dispatch returned %d\n", rv);   // stm32f103/synth.c:36
    return rv;
}

extern "C" int smx_fw_tick_n(smx_fw *f, uint32_t n_ticks, const uint32_t *dither, uint8_t *duty)
{
    if (!f) return SMX_E_ARG;
    if (!f->running) return 0;
    const int rv = smx_pwm_tick_n(f->pwm, n_ticks, dither, duty);
    return rv ? rv : (int)n_ticks;
}

extern "C" int smx_fw_poll(smx_fw *f, uint32_t osc, uint32_t *avg, uint32_t *num,
                           uint8_t *cont, uint32_t cont_cap, uint32_t *cont_len)
{
    if (!f || !f->osc || osc >= f->read_idx.size()) return SMX_E_ARG;
    if (cont_len) *cont_len = 0;
    smx_osc *o = f->osc;
    SMX_HIP(hipSetDevice(o->device));
    SMX_HIP(hipStreamSynchronize(o->stream));
    uint32_t write = 0;
    SMX_HIP(hipMemcpy(&write, o->pm.write + osc, 4, hipMemcpyDeviceToHost));
    if (f->read_idx[osc] == write) return 0;                 // pmeas.h:32
    const uint32_t read = ++f->read_idx[osc];                // pmeas.h:35
    uint32_t a = 0, n = 0;
    SMX_HIP(hipMemcpy(&a, ((read & 1) ? o->pm.avg1 : o->pm.avg0) + osc, 4, hipMemcpyDeviceToHost));
    SMX_HIP(hipMemcpy(&n, ((read & 1) ? o->pm.num1 : o->pm.num0) + osc, 4, hipMemcpyDeviceToHost));
    if (avg) *avg = a;
    if (num) *num = n;
    if (!f->wait.empty()) {                                  // pmeas.h:44-58
        const uint32_t len = f->wait[0];
        if (len + 1 > f->wait.size()) {
            f->wait.clear();                                 // "bad measurement_wait size ... clearing"
        } else {
            if (cont && len <= cont_cap) memcpy(cont, f->wait.data() + 1, len);
            if (cont_len) *cont_len = len;
            f->wait.erase(f->wait.begin(), f->wait.begin() + 1 + len);
        }
    }
    return 1;
}

