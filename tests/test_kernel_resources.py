"""Register and LDS budgets of the kernels of the headline pass, read from the code objects inside libffk.so
(CPU test: no GPU, no external tool -- the offload bundles and the AMDGPU metadata notes are parsed here).

Why this is a test (profiles/r05_j_*): the bench's two passes overlap only while one
wavefront of a pass's small kernels fits beside three wavefronts of the other pass's accumulate kernel on a
SIMD: 3 x 152 + 56 <= 512 vector registers, static LDS of the small kernels <= 8 KiB, blocks of <= 256
threads.  A refactor that made hipcc call the eigensolver's wave function instead of inlining it took the
stand-alone eigensolver from 52 to 120 registers and the headline step from 60 to 73 us without a single
numerical difference."""
import os
import struct

import pytest

from conftest import ROOT

msgpack = pytest.importorskip('msgpack')

LIB = os.path.join(ROOT, 'filter_functions_amd', 'libffk.so')
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def _elf_sections(blob):
    """(name, type, offset, size) of the sections of a 64-bit little-endian ELF image"""
    assert blob[:4] == b'\x7fELF' and blob[4] == 2 and blob[5] == 1
    shoff, = struct.unpack_from('<Q', blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from('<HHH', blob, 0x3A)
    heads = [struct.unpack_from('<IIQQQQIIQQ', blob, shoff + i*shentsize) for i in range(shnum)]
    names = heads[shstrndx]
    strtab = blob[names[4]:names[4] + names[5]]
    out = []
    for h in heads:
        end = strtab.index(b'\0', h[0])
        out.append((strtab[h[0]:end].decode(), h[1], h[4], h[5]))
    return out


def _device_images(lib_path, arch='gfx950'):
    blob = open(lib_path, 'rb').read()
    fat = [(off, size) for name, _, off, size in _elf_sections(blob) if name == '.hip_fatbin']
    assert fat, 'no .hip_fatbin section in libffk.so'
    off, size = fat[0]
    section = blob[off:off + size]
    pos = section.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from('<Q', section, pos + len(MAGIC))
        cursor = pos + len(MAGIC) + 8
        for _ in range(n):
            eoff, esize, tsize = struct.unpack_from('<QQQ', section, cursor)
            triple = section[cursor + 24:cursor + 24 + tsize].decode()
            cursor += 24 + tsize
            if arch in triple and esize:
                yield section[pos + eoff:pos + eoff + esize]
        pos = section.find(MAGIC, pos + len(MAGIC))


def _kernels(image):
    for name, stype, off, size in _elf_sections(image):
        if stype != 7:          # SHT_NOTE
            continue
        cursor = off
        while cursor < off + size:
            namesz, descsz, ntype = struct.unpack_from('<III', image, cursor)
            cursor += 12
            owner = image[cursor:cursor + namesz].rstrip(b'\0')
            cursor += (namesz + 3) & ~3
            desc = image[cursor:cursor + descsz]
            cursor += (descsz + 3) & ~3
            if owner == b'AMDGPU' and ntype == 32:
                meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                yield from meta.get('amdhsa.kernels', [])


@pytest.fixture(scope='module')
def kernels():
    if not os.path.exists(LIB):
        pytest.skip('libffk.so not built')
    found = {}
    for image in _device_images(LIB):
        for k in _kernels(image):
            found[k['.name']] = k
    assert len(found) > 50, f'only {len(found)} kernels found in the library'
    return found


def _one(kernels, *fragments):
    hits = [k for name, k in kernels.items() if all(f in name for f in fragments)]
    assert len(hits) == 1, (fragments, [k['.name'] for k in hits])
    return hits[0]


def test_small_kernels_of_the_headline_pass_fit_beside_the_accumulate_kernel(kernels):
    small = [_one(kernels, 'eigh_expm_kernelILi4E'), _one(kernels, 'scan_local_kernelILi4E'),
             _one(kernels, 'apply_prologue_kernelILi4E'), _one(kernels, 'expand_ff_kernel'),
             _one(kernels, 'infid_kernelILb0E')]
    for k in small:
        assert k['.vgpr_count'] <= 56, (k['.name'], k['.vgpr_count'])
        assert k['.vgpr_spill_count'] == 0 and k['.sgpr_spill_count'] == 0, k['.name']
        assert k['.group_segment_fixed_size'] <= 8192, (k['.name'], k['.group_segment_fixed_size'])
        assert k['.max_flat_workgroup_size'] <= 256, (k['.name'], k['.max_flat_workgroup_size'])


def test_d4_accumulate_kernel_leaves_room_for_a_second_pass(kernels):
    for nc in (1, 2, 3):
        for pre in (0, 1):          # W_a folded in the kernel / by the prologue kernel
            k = _one(kernels, 'ctrl_accumulate_pq_kernelILi%dELb%dE' % (nc, pre))
            # three wavefronts per SIMD at <= 152 registers (allocation granule 8) + one small-kernel wavefront at 56
            assert k['.vgpr_count'] <= 152, (nc, pre, k['.vgpr_count'])
            assert k['.vgpr_spill_count'] == 0 and k['.sgpr_spill_count'] == 0, (nc, pre)
            assert k['.max_flat_workgroup_size'] == 768


def test_matrix_core_accumulate_kernels_keep_their_occupancy(kernels):
    """d = 8 (config 4): 16 wavefronts per block = four per SIMD -> at most 128 registers; d = 16 (config 5): eight
    wavefronts = two per SIMD -> at most 256."""
    for pre in (0, 1):              # W' folded by the kernel's producers / copied from the prologue's fold by LDS-DMA
        k8 = _one(kernels, 'ctrl_accumulate_pcr_kernelILi3ELb%dE' % pre)
        assert k8['.vgpr_count'] <= 128 and k8['.max_flat_workgroup_size'] == 1024, k8['.vgpr_count']
    k16 = _one(kernels, 'ctrl_accumulate_mfma4_kernelILi16ELi2ELi8ELb1E')      # the instantiation config 5 launches
    assert k16['.vgpr_count'] <= 256 and k16['.max_flat_workgroup_size'] == 512, k16['.vgpr_count']


def test_hot_kernels_keep_nothing_in_private_memory(kernels):
    """A value that the compiler keeps in scratch memory although nothing is spilled -- a struct assigned under a
    lane-type branch, a struct copy held across a barrier -- turns every load that feeds it into `global_load;
    s_waitcnt vmcnt(0); scratch_store`: the first build of the d = 2 kernel spent 4 us per segment that way, and the
    general accumulate kernel and the prologue had the same pattern (profiles/r06_g_*, last sections).  The kernels of
    the measured paths use no private memory at all (the d = 8 kernel: one spilled register pair outside its loops)."""
    clean = [name for name in kernels
             if any(f in name for f in ('ctrl_accumulate_d2_kernel', 'ctrl_accumulate_pq_kernelILi3ELb1E',
                                        'decay_gemm_sym256_kernel', 'conjugate_basis_mfma_kernelILi16ELb1ELb1E',
                                        'ctrl_accumulate_mfma4_kernelILi16ELi1ELi4ELb1ELi8E', 'expand_ff_kernel',
                                        'eigh_expm_kernelILi4E', 'scan_local_kernelILi4E', 'infid_kernelILb0E'))]
    clean += [name for name in kernels if 'apply_prologue_kernelILi' in name and 'ILi16E' not in name]
    clean += [name for name in kernels if 'ctrl_accumulate_kernelILi' in name and
              any('kernelILi%dE' % d in name for d in (2, 3, 6, 7, 9, 10, 11))]
    assert len(clean) > 100, len(clean)
    for name in clean:
        assert kernels[name]['.private_segment_fixed_size'] == 0, (name, kernels[name]['.private_segment_fixed_size'])
    k8 = _one(kernels, 'ctrl_accumulate_pcr_kernelILi3ELb1E')
    assert k8['.private_segment_fixed_size'] <= 16 and k8['.vgpr_spill_count'] <= 2

