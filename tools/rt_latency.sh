#!/bin/bash
# tools/rt_latency.sh -- wall time of the JACK process() callback (fake-JACK harness) per 64-frame block
# for growing banks, synchronous and pipelined.  Run on the GPU box from the repo root.
: > /tmp/ev.bin
# every voice of the bank sounding (SYNTH_FILL: piano-range notes), paced at the JACK operating point (64 frames @ 48 kHz = 1333 us per callback)
for v in 64 65536 1048576 16777216 67108864 268435456; do
  SYNTH_FILL=1 SYNTH_FAKE_PERIOD_US=1333 SYNTH_VOICES=$v ./host/synth.dynamic.host.elf --fake-jack 300 64 /tmp/ev.bin /tmp/out.f32 < /dev/null 2>&1 | grep fake-jack | sed "s/^/sync      /"
  SYNTH_FILL=1 SYNTH_FAKE_PERIOD_US=1333 SYNTH_VOICES=$v SYNTH_PIPELINE=1 ./host/synth.dynamic.host.elf --fake-jack 300 64 /tmp/ev.bin /tmp/out.f32 < /dev/null 2>&1 | grep fake-jack | sed "s/^/pipelined /"
done
# worst case for the wrap-event form: every voice at 12 wraps per block.  Default (AUTO): the device's statistic keeps
# the stepping form on this bank, so the callback takes the stepping form's time; SYNTH_FORM_STEPPING=1 pins that form.
for v in 16777216 67108864 268435456; do
  SYNTH_FILL=worst SYNTH_FAKE_PERIOD_US=1333 SYNTH_VOICES=$v ./host/synth.dynamic.host.elf --fake-jack 300 64 /tmp/ev.bin /tmp/out.f32 < /dev/null 2>&1 | grep fake-jack | sed "s/^/sync worst-case bank, default (AUTO)        /"
  SYNTH_FILL=worst SYNTH_FORM_STEPPING=1 SYNTH_FAKE_PERIOD_US=1333 SYNTH_VOICES=$v ./host/synth.dynamic.host.elf --fake-jack 300 64 /tmp/ev.bin /tmp/out.f32 < /dev/null 2>&1 | grep fake-jack | sed "s/^/sync worst-case bank, SYNTH_FORM_STEPPING=1  /"
  SYNTH_FILL=1 SYNTH_FORM_STEPPING=1 SYNTH_FAKE_PERIOD_US=1333 SYNTH_VOICES=$v ./host/synth.dynamic.host.elf --fake-jack 300 64 /tmp/ev.bin /tmp/out.f32 < /dev/null 2>&1 | grep fake-jack | sed "s/^/sync piano-range bank, SYNTH_FORM_STEPPING=1 /"
done
# other JACK block lengths of the 64 Mi-voice bank (round 3: one 32-frame chunk / one 128-frame chunk of the event form)
for fr in 32 128 256; do
  SYNTH_FILL=1 SYNTH_FAKE_PERIOD_US=$((fr * 1000000 / 48000)) SYNTH_VOICES=67108864 ./host/synth.dynamic.host.elf --fake-jack 300 $fr /tmp/ev.bin /tmp/out.f32 < /dev/null 2>&1 | grep fake-jack | sed "s/^/sync, $fr-frame blocks          /"
  SYNTH_FILL=1 SYNTH_FORM_STEPPING=1 SYNTH_FAKE_PERIOD_US=$((fr * 1000000 / 48000)) SYNTH_VOICES=67108864 ./host/synth.dynamic.host.elf --fake-jack 300 $fr /tmp/ev.bin /tmp/out.f32 < /dev/null 2>&1 | grep fake-jack | sed "s/^/sync, $fr-frame blocks, SYNTH_FORM_STEPPING=1 /"
done
