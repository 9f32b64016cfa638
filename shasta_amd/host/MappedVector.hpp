// Shasta's memory-mapped containers, as files: the literal data contract of the two seams.
//
// A Shasta run directory keeps every large structure as a file under Data/ (SURVEY 8b):
//   MemoryMapped::Vector<T>            src/MemoryMappedVector.hpp:165-231: a 4096-byte header
//       {headerSize, objectSize, objectCount, pageSize, pageCount, fileSize, capacity,
//        magicNumber = 0xa3756fd4b5d8bcc1, zero padding}, then objectCount objects of objectSize
//       bytes; the file is a whole number of pages.  Opening checks the magic number, that
//       fileSize is the size of the file and that objectSize is sizeof(T) (:624-642).
//   MemoryMapped::VectorOfVectors<T,Int>  src/MemoryMappedVectorOfVectors.hpp:28-42: "<name>.toc"
//       = Vector<Int> of n+1 offsets and "<name>.data" = Vector<T>.
// These classes read and write exactly that layout so that a Data/ directory written by Shasta
// is the input of this host layer and what this layer writes is what Shasta's next stage opens.
// They are written from the format, not from the reference's code: plain files, one mmap each,
// growth by ftruncate + mremap.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace shasta_mi355x {
namespace host {

struct MappedHeader {
    uint64_t headerSize, objectSize, objectCount, pageSize, pageCount, fileSize, capacity, magicNumber;
    uint64_t padding[4096 / 8 - 8];
};
static_assert(sizeof(MappedHeader) == 4096, "MemoryMapped::Vector header is 4096 bytes");
constexpr uint64_t MAPPED_MAGIC = 0xa3756fd4b5d8bcc1ULL;

template<class T> class MappedVector {
public:
    MappedVector() = default;
    MappedVector(const MappedVector&) = delete;
    MappedVector& operator=(const MappedVector&) = delete;
    ~MappedVector() { close(); }

    // A new, empty vector backed by file `name` (page size 4096 or 2 MiB as in Shasta).
    void createNew(const std::string& name, uint64_t pageSize = 4096)
    {
        close();
        fileName = name; writable = true;
        const int fd = ::open(name.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
        if(fd < 0) throw std::runtime_error("Error creating " + name);
        const uint64_t bytes = pageSize * pagesFor(sizeof(MappedHeader), pageSize);
        if(::ftruncate(fd, off_t(bytes)) != 0) { ::close(fd); throw std::runtime_error("Error sizing " + name); }
        map(fd, bytes);
        ::close(fd);
        std::memset(header, 0, sizeof(MappedHeader));
        header->headerSize = sizeof(MappedHeader); header->objectSize = sizeof(T); header->objectCount = 0;
        header->pageSize = pageSize; header->pageCount = bytes / pageSize; header->fileSize = bytes;
        header->capacity = (bytes - sizeof(MappedHeader)) / sizeof(T); header->magicNumber = MAPPED_MAGIC;
    }

    // Same checks and messages as MemoryMapped::Vector::accessExisting (:624-642).
    void accessExistingReadOnly(const std::string& name) { accessExisting(name, false); }
    void accessExistingReadWrite(const std::string& name) { accessExisting(name, true); }

private:
    void accessExisting(const std::string& name, bool readWrite)
    {
        close();
        fileName = name; writable = readWrite;
        const int fd = ::open(name.c_str(), readWrite ? O_RDWR : O_RDONLY);
        if(fd < 0) throw std::runtime_error("Error accessing " + name + ": the file could not be opened.");
        struct stat st;
        if(::fstat(fd, &st) != 0 || uint64_t(st.st_size) < sizeof(MappedHeader)) { ::close(fd); throw std::runtime_error("Error accessing " + name + ": file too small."); }
        map(fd, uint64_t(st.st_size));
        ::close(fd);
        if(header->magicNumber != MAPPED_MAGIC) { const std::string n = name; close(); throw std::runtime_error("Error accessing " + n + ": unexpected magic number in header. The binary format of this file is not recognized. Perhaps a file mixup?"); }
        if(header->fileSize != mappedBytes) { const std::string n = name; close(); throw std::runtime_error("Error accessing " + n + ": file size not consistent with file header. Perhaps a file mixup?"); }
        if(header->objectSize != sizeof(T)) {
            const std::string n = name; const uint64_t found = header->objectSize; close();
            throw std::runtime_error("Error accessing " + n + ": unexpected object size. Expected " + std::to_string(sizeof(T)) + ", found " + std::to_string(found) + ".");
        }
    }

public:
    bool isOpen() const { return header != nullptr; }
    uint64_t size() const { return header ? header->objectCount : 0; }
    uint64_t capacity() const { return header ? header->capacity : 0; }
    T* begin() { return dataPointer; }
    const T* begin() const { return dataPointer; }
    T* end() { return dataPointer + size(); }
    const T* end() const { return dataPointer + size(); }
    T& operator[](uint64_t i) { return dataPointer[i]; }
    const T& operator[](uint64_t i) const { return dataPointer[i]; }

    void reserve(uint64_t n)
    {
        requireWritable();
        if(n <= header->capacity) return;
        remap(n);
    }
    void resize(uint64_t n)        // new objects are zero bytes
    {
        requireWritable();
        if(n > header->capacity) remap(n + n / 2);
        if(n > header->objectCount) std::memset(static_cast<void*>(dataPointer + header->objectCount), 0, (n - header->objectCount) * sizeof(T));
        header->objectCount = n;
    }
    void push_back(const T& t)
    {
        requireWritable();
        if(header->objectCount == header->capacity) remap(header->objectCount + header->objectCount / 2 + 1024);
        std::memcpy(static_cast<void*>(dataPointer + header->objectCount), &t, sizeof(T));
        ++header->objectCount;
    }
    void append(const T* first, uint64_t n)
    {
        requireWritable();
        if(header->objectCount + n > header->capacity) remap(header->objectCount + n);
        if(n) std::memcpy(static_cast<void*>(dataPointer + header->objectCount), first, n * sizeof(T));
        header->objectCount += n;
    }
    // Shrinks the file to the pages needed (MemoryMapped::Vector::unreserve).
    void unreserve() { requireWritable(); remap(header->objectCount); }

    void close()
    {
        if(header) { ::munmap(header, mappedBytes); header = nullptr; dataPointer = nullptr; mappedBytes = 0; }
    }
    void remove() { const std::string n = fileName; close(); if(!n.empty()) ::unlink(n.c_str()); }

private:
    static uint64_t pagesFor(uint64_t bytes, uint64_t pageSize) { return (bytes - 1) / pageSize + 1; }
    void requireWritable() const { if(!header || !writable) throw std::runtime_error("MappedVector " + fileName + " is not open for writing."); }
    void map(int fd, uint64_t bytes)
    {
        void* p = ::mmap(nullptr, bytes, writable ? (PROT_READ | PROT_WRITE) : PROT_READ, MAP_SHARED, fd, 0);
        if(p == MAP_FAILED) throw std::runtime_error("Error mapping " + fileName);
        header = static_cast<MappedHeader*>(p);
        dataPointer = reinterpret_cast<T*>(header + 1);
        mappedBytes = bytes;
    }
    void remap(uint64_t newCapacity)
    {
        const uint64_t pageSize = header->pageSize;
        const uint64_t bytes = pageSize * pagesFor(sizeof(MappedHeader) + sizeof(T) * newCapacity, pageSize);
        if(bytes != mappedBytes) {
            const int fd = ::open(fileName.c_str(), O_RDWR);
            if(fd < 0) throw std::runtime_error("Error reopening " + fileName);
            ::munmap(header, mappedBytes);
            header = nullptr;
            if(::ftruncate(fd, off_t(bytes)) != 0) { ::close(fd); throw std::runtime_error("Error resizing " + fileName); }
            map(fd, bytes);
            ::close(fd);
        }
        header->pageCount = bytes / pageSize; header->fileSize = bytes;
        header->capacity = (bytes - sizeof(MappedHeader)) / sizeof(T);
    }

    MappedHeader* header = nullptr;
    T* dataPointer = nullptr;
    uint64_t mappedBytes = 0;
    bool writable = false;
    std::string fileName;
};

template<class T, class Int> class MappedVectorOfVectors {
public:
    void createNew(const std::string& name, uint64_t pageSize = 4096)
    {
        toc.createNew(name + ".toc", pageSize); data.createNew(name + ".data", pageSize);
        toc.push_back(Int(0));
    }
    void accessExistingReadOnly(const std::string& name)
    {
        toc.accessExistingReadOnly(name + ".toc"); data.accessExistingReadOnly(name + ".data");
        if(toc.size() == 0 || uint64_t(toc[toc.size() - 1]) != data.size()) throw std::runtime_error("Error accessing " + name + ": table of contents not consistent with data.");
    }
    uint64_t size() const { return toc.size() ? toc.size() - 1 : 0; }       // number of vectors
    uint64_t size(uint64_t i) const { return uint64_t(toc[i + 1]) - uint64_t(toc[i]); }
    uint64_t totalSize() const { return data.size(); }
    const T* begin(uint64_t i) const { return data.begin() + uint64_t(toc[i]); }
    T* begin(uint64_t i) { return data.begin() + uint64_t(toc[i]); }
    void appendVector(const T* first, uint64_t n) { data.append(first, n); toc.push_back(Int(data.size())); }
    // Two-pass fill with known counts (the shape of beginPass1 / beginPass2 / store, src/MemoryMappedVectorOfVectors.hpp:315-393).
    void fillFromCounts(const std::vector<Int>& counts)
    {
        toc.resize(counts.size() + 1);
        uint64_t s = 0;
        toc[0] = Int(0);
        for(size_t i = 0; i < counts.size(); i++) { s += uint64_t(counts[i]); toc[i + 1] = Int(s); }
        data.resize(s);
    }
    void unreserve() { toc.unreserve(); data.unreserve(); }
    void close() { toc.close(); data.close(); }
    MappedVector<Int> toc;
    MappedVector<T> data;
};

}  // namespace host
}  // namespace shasta_mi355x
